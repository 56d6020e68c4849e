#!/bin/bash
# Round-end measurement on the GPU box (run through gpurun): default bench line, rocprofv3 kernel stats of
# the same command, and the two HBM-traffic PMC passes (FETCH_SIZE, WRITE_SIZE in separate runs) over the per-block
# leg alone (every launch there is one step's 16384 long blocks, so a kernel's median is a per-launch figure).
# Outputs land in gpurun_out/final/; tools/collect_profiles.py turns them into profiles/.
set -o pipefail
OUT=$GRAFT_REPO_ROOT/gpurun_out/final
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $OUT/stats.log 2>&1; echo "stats rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o f -- python3 $GRAFT_REPO_ROOT/bench.py --only block --steps 6 --warmup 2 --no-cpu-baseline > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o w -- python3 $GRAFT_REPO_ROOT/bench.py --only block --steps 6 --warmup 2 --no-cpu-baseline > $OUT/write.log 2>&1; echo "write rc=$?"
cd $GRAFT_REPO_ROOT
python3 tools/collect_profiles.py ${1:-r02} $OUT/profiles $2 && rm -rf $OUT/stats $OUT/fetch $OUT/write
ls -la $OUT $OUT/profiles/*
cut -c1-600 $OUT/bench.json
