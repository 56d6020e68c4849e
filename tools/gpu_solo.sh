#!/bin/bash
# Solo kernel times of the per-block path: one HIP stream, mask branches in sequence (no kernel runs beside another).
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/solo
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VBM_OVERLAP_BRANCHES=0 VBM_BENCH_TWO_STREAMS=0
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --only block --steps 12 --warmup 3 --no-cpu-baseline > $OUT/stats.log 2>&1; echo "stats rc=$?"
python3 $GRAFT_REPO_ROOT/tools/kstats.py $(ls $OUT/stats/*/*.db $OUT/stats/*.db 2>/dev/null | head -1) > $OUT/kstats.txt 2>&1
head -40 $OUT/kstats.txt
tail -1 $OUT/stats.log | cut -c1-300
rm -rf $OUT/stats
