#!/bin/bash
# kernel statistics of the drop-in benchmark (deferred delivery)
export LD_LIBRARY_PATH=$GRAFT_REPO_ROOT/vorbis_aotuv_lancer_amd:$LD_LIBRARY_PATH
OUT=$GRAFT_REPO_ROOT/gpurun_out/compattrace
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
VORBIS_MI355X_DEFER_BLOCKS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- $GRAFT_REPO_ROOT/examples/compat_bench 4 4096 4096 24 8 > $OUT/run.log 2>&1; echo "rc=$?"
DB=$(ls $OUT/stats/*/*.db $OUT/stats/*.db 2>/dev/null | head -1)
python3 $GRAFT_REPO_ROOT/tools/kstats.py $DB > $OUT/kstats.txt 2>&1
head -40 $OUT/kstats.txt
tail -2 $OUT/run.log | cut -c1-300
rm -rf $OUT/stats
