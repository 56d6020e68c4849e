#!/bin/bash
# Issue work per kernel of the per-block path (solo leg): VALU / SALU / LDS instruction counts and wave counts.
# usage: tools/gpu_pmc_work.sh [ENV=VALUE ...]
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/work
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for kv in "$@"; do export "$kv"; done
export VBM_BENCH_SOLO_STEPS=4
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES --kernel-trace -d $OUT/a -o a -- python3 $GRAFT_REPO_ROOT/bench.py --only solo > $OUT/a.log 2>&1; echo "rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVE_CYCLES --kernel-trace -d $OUT/b -o b -- python3 $GRAFT_REPO_ROOT/bench.py --only solo > $OUT/b.log 2>&1; echo "rc=$?"
python3 $GRAFT_REPO_ROOT/tools/pmcstats.py $(ls $OUT/a/*/*.db $OUT/a/*.db 2>/dev/null | head -1) $(ls $OUT/b/*/*.db $OUT/b/*.db 2>/dev/null | head -1) > $OUT/work.txt 2>&1
cat $OUT/work.txt
rm -rf $OUT/a $OUT/b
