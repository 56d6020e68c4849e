#!/bin/bash
# timing experiment: the fused noise-mask kernel with phases switched off (results are wrong then; only the time counts)
cd $GRAFT_REPO_ROOT
export VBM_OVERLAP_BRANCHES=0 VBM_BENCH_TWO_STREAMS=0
for nbk in ${NBS:-4}; do
for ph in ${PHASES:-0 1 2 3 7 15 16 31}; do
  VBM_NOISE_NB=$nbk VBM_NOISE_PHASES=$ph python3 bench.py --only block --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('nb', $nbk, 'phases', $ph, 'noisemask ms', round(d['stage_ms_per_step']['noisemask'],4))"
done; done
