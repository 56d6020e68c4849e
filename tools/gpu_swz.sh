#!/bin/bash
set -e
python3 -m pytest tests/test_pipeline_gpu.py tests/test_frontend_gpu.py tests/test_reference_input_gpu.py -x -q -m gpu > gpurun_out/swz_tests.log 2>&1 || { tail -30 gpurun_out/swz_tests.log; exit 1; }
tail -3 gpurun_out/swz_tests.log
for v in 0 1; do
  echo -n "solo VBM_NOISE_SWZ=$v : "
  VBM_NOISE_SWZ=$v python3 bench.py --only solo --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['stage_solo_ms']['noisemask'],3))"
done
bash tools/gpu_ab_vals.sh pcm VBM_NOISE_SWZ 0 1
bash tools/gpu_ab_vals.sh block VBM_NOISE_SWZ 0 1
