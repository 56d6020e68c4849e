#!/bin/bash
set -e
python3 -m pytest tests/test_fft_gpu.py tests/test_pipeline_gpu.py -x -q -m gpu > gpurun_out/fft_tests.log 2>&1 || { tail -30 gpurun_out/fft_tests.log; exit 1; }
tail -2 gpurun_out/fft_tests.log
python3 bench.py --only solo --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print({k: round(v,3) for k,v in d['stage_solo_ms'].items()})"
bash tools/gpu_repeat.sh pcm 4 VBM_X=0
bash tools/gpu_repeat.sh block 3 VBM_X=0
