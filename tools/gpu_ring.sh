#!/bin/bash
# ring form of the noise mask kernel: parity first, then solo and in-pipeline timings with the switch off / on
set -e
python3 -m pytest tests/test_pipeline_gpu.py tests/test_frontend_gpu.py -x -q -m gpu > gpurun_out/ring_tests.log 2>&1 || { tail -30 gpurun_out/ring_tests.log; exit 1; }
tail -3 gpurun_out/ring_tests.log
for v in 0 1; do
  echo "solo VBM_NOISE_RING=$v"
  VBM_NOISE_RING=$v python3 bench.py --only solo --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k: round(v,3) for k,v in d['stage_solo_ms'].items()})"
done
bash tools/gpu_ab_vals.sh pcm VBM_NOISE_RING 0 1
bash tools/gpu_ab_vals.sh block VBM_NOISE_RING 0 1
