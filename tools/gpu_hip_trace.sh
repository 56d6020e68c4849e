#!/bin/bash
# Host API timeline beside the kernel timeline of the from-PCM leg (who waits for whom: host or device)
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/hiptrace
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VBM_BENCH_NO_STAGE_PASS=1
timeout -k 10 400 rocprofv3 --kernel-trace --hip-runtime-trace -d $OUT/tr -o t -- python3 $GRAFT_REPO_ROOT/bench.py --only pcm --steps ${STEPS:-8} --warmup 4 --no-cpu-baseline > $OUT/run.log 2>&1; echo "rc=$?"
DB=$(ls $OUT/tr/*/*.db $OUT/tr/*.db 2>/dev/null | head -1)
python3 $GRAFT_REPO_ROOT/tools/hip_timeline.py $DB ${LAST_MS:-14} > $OUT/timeline.txt 2>&1
tail -3 $OUT/timeline.txt
rm -rf $OUT/tr
