#!/bin/bash
# Kernel trace of the from-PCM leg: per-kernel stats and the timeline of the last dispatches.
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/pcmtrace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VBM_BENCH_NO_STAGE_PASS=1
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py ${BENCH_EXTRA} --only pcm --steps ${STEPS:-12} --warmup 4 --no-cpu-baseline > $OUT/stats.log 2>&1; echo "stats rc=$?"
DB=$(ls $OUT/stats/*/*.db $OUT/stats/*.db 2>/dev/null | head -1)
python3 $GRAFT_REPO_ROOT/tools/kstats.py $DB > $OUT/kstats.txt 2>&1
python3 $GRAFT_REPO_ROOT/tools/ktimeline.py $DB ${LAST:-700} > $OUT/timeline.txt 2>&1
head -50 $OUT/kstats.txt
tail -1 $OUT/stats.log | cut -c1-400
rm -rf $OUT/stats
