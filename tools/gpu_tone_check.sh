#!/bin/bash
cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_pipeline_gpu.py -x -q 2>&1 | tail -3
export VBM_OVERLAP_BRANCHES=0 VBM_BENCH_TWO_STREAMS=0
for ri in 16 32 48; do
for ph in 4 31; do
  VBM_TONE_RUNIN=$ri VBM_TONE_PHASES=$ph python3 bench.py --only block --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('runin', $ri, 'phases', $ph, 'tonemask ms', d['stage_ms_per_step']['tonemask'])"
done
done
