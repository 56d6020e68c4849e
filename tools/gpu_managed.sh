#!/bin/bash
# managed bitrate: parity, then the two legs of the bench
python3 -m pytest tests/test_managed_gpu.py tests/test_frontend_gpu.py tests/test_compat_gpu.py -x -q -m gpu > gpurun_out/managed_tests.log 2>&1 || { tail -40 gpurun_out/managed_tests.log; exit 1; }
tail -3 gpurun_out/managed_tests.log
for leg in block pcm; do for w in ${WIDE:-1}; do
  echo -n "$leg VBM_MANAGED_WIDE=$w  "
  VBM_MANAGED_WIDE=$w python3 bench.py --bitrate 128000 --only $leg --steps 24 --warmup 4 --no-cpu-baseline 2>gpurun_out/managed_err_$leg$w.log | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step'],3), round(d['value']), d['config'].get('short_block_fraction'), d['config'].get('encoded_over_input'))"
done; done
