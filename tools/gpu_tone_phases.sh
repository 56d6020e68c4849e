#!/bin/bash
cd $GRAFT_REPO_ROOT
export VBM_OVERLAP_BRANCHES=0 VBM_BENCH_TWO_STREAMS=0
for ph in 0 1 2 4 8 16 31; do
  VBM_TONE_PHASES=$ph python3 bench.py --only block --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('phases', $ph, 'tonemask ms', d['stage_ms_per_step']['tonemask'])"
done
