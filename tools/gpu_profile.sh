#!/bin/bash
# Round-end measurement on the GPU box (run through gpurun): tools/gpu_profile.sh <round> <out-dir-under-gpurun_out> <commit>
#   1. the default bench line (what the driver runs)                                   -> bench_default.json
#      and the managed-bitrate line (bench.py --bitrate 128000)                        -> bench_bitrate128k.json
#   2. rocprofv3 --kernel-trace --stats of the same command                            -> kernel_stats.csv
#   3. rocprofv3 --kernel-trace --stats of `bench.py --only solo`: every stage of the per-block path alone, full-size
#      launches only (what the bench line's roofline is made of)                       -> solo_kernel_stats.csv
#   4. HBM traffic: FETCH_SIZE / WRITE_SIZE in separate PMC passes over the per-block leg (per launch and kernel) and over
#      the from-PCM leg (per write, all kernels incl. the front end)                   -> pmc_hbm_traffic*.csv, pmc_traffic.json
#   5. issue work: SQ_INSTS_VALU / SALU / LDS per kernel, solo leg and from-PCM leg    -> valu_work_*.txt
# Databases are too large to travel back: tools/collect_profiles.py reduces them here.
set -o pipefail
RND=${1:-r03}
OUT=$GRAFT_REPO_ROOT/gpurun_out/final
rm -rf $OUT; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$?"
timeout -k 10 400 python3 bench.py --bitrate 128000 --steps 20 --warmup 5 > $OUT/bench_bitrate128k.json 2> $OUT/bench_bitrate.err; echo "bench --bitrate rc=$?"
cd /tmp && export TMPDIR=/tmp
B=$GRAFT_REPO_ROOT/bench.py
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o s -- python3 $B --steps 20 --warmup 5 --no-cpu-baseline > $OUT/stats.log 2>&1; echo "stats rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/solo -o s -- python3 $B --only solo > $OUT/solo.log 2>&1; echo "solo rc=$?"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetch -o f -- python3 $B --only block --steps 6 --warmup 2 --no-cpu-baseline > $OUT/fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/write -o w -- python3 $B --only block --steps 6 --warmup 2 --no-cpu-baseline > $OUT/write.log 2>&1; echo "write rc=$?"
export VBM_BENCH_NO_STAGE_PASS=1 VBM_BENCH_PRIME=16
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $OUT/fetchp -o f -- python3 $B --only pcm --steps 24 --warmup 4 --no-cpu-baseline > $OUT/fetchp.log 2>&1; echo "fetch pcm rc=$?"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $OUT/writep -o w -- python3 $B --only pcm --steps 24 --warmup 4 --no-cpu-baseline > $OUT/writep.log 2>&1; echo "write pcm rc=$?"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace -d $OUT/workp -o q -- python3 $B --only pcm --steps 24 --warmup 4 --no-cpu-baseline > $OUT/workp.log 2>&1; echo "work pcm rc=$?"
unset VBM_BENCH_NO_STAGE_PASS VBM_BENCH_PRIME
export VBM_BENCH_SOLO_STEPS=4
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES --kernel-trace -d $OUT/works -o q -- python3 $B --only solo > $OUT/works.log 2>&1; echo "work solo rc=$?"
cd $GRAFT_REPO_ROOT
python3 tools/collect_profiles.py $RND $OUT/profiles $2 && rm -rf $OUT/stats $OUT/solo $OUT/fetch $OUT/write $OUT/fetchp $OUT/writep $OUT/workp $OUT/works
cp $OUT/bench_bitrate128k.json $OUT/profiles/bench_bitrate128k.json
ls -la $OUT/profiles $OUT/profiles/*
cut -c1-400 $OUT/bench.json
