#!/bin/bash
# vector instructions of the tone / noise mask kernels with phases switched off (VBM_TONE_PHASES / VBM_NOISE_PHASES bits)
OUT=$GRAFT_REPO_ROOT/gpurun_out/phasevalu
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export VBM_BENCH_SOLO_STEPS=2
for ph in 31 30 29 27 23 15; do
  export VBM_TONE_PHASES=$ph
  timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -d $OUT/p$ph -o q -- python3 $GRAFT_REPO_ROOT/bench.py --only solo > $OUT/p$ph.log 2>&1
  echo -n "tone phases=$ph: "; python3 $GRAFT_REPO_ROOT/tools/pmc_kernel_sum.py $OUT/p$ph k_tonemask
  rm -rf $OUT/p$ph
done
