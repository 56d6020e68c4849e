#!/bin/bash
# instruction counts of the kernels matching a name, solo leg: tools/gpu_pmc_kernel.sh SUBSTR [ENV=VAL ...]
K=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/work
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
export VBM_BENCH_SOLO_STEPS=2
rm -rf $OUT/p; mkdir -p $OUT/p
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace -d $OUT/p -o p -- python3 $GRAFT_REPO_ROOT/bench.py --only solo > $OUT/p.log 2>&1
python3 $GRAFT_REPO_ROOT/tools/pmcstats.py $(ls $OUT/p/*/*.db $OUT/p/*.db 2>/dev/null | head -1) --filter $K
rm -rf $OUT/p
