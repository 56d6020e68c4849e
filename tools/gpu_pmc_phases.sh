#!/bin/bash
# VALU / LDS / SALU instruction counts of one kernel with phases switched off (timing-experiment switches):
#   tools/gpu_pmc_phases.sh KERNEL_SUBSTR ENVVAR v1 v2 ...
K=$1; VAR=$2; shift 2
OUT=$GRAFT_REPO_ROOT/gpurun_out/work
cd /tmp && export TMPDIR=/tmp
export VBM_BENCH_SOLO_STEPS=2
for v in "$@"; do
  rm -rf $OUT/p; mkdir -p $OUT/p
  env $VAR=$v timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace -d $OUT/p -o p -- python3 $GRAFT_REPO_ROOT/bench.py --only solo > $OUT/p.log 2>&1
  echo -n "$VAR=$v  "; python3 $GRAFT_REPO_ROOT/tools/pmcstats.py $(ls $OUT/p/*/*.db $OUT/p/*.db 2>/dev/null | head -1) --filter $K | tail -1
done
rm -rf $OUT/p
