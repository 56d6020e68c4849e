#!/bin/bash
# how much of a from-PCM step is the host enqueueing it?
for setting in "VBM_X=0" "VBM_DEVICE_GRAPHS=0" "VBM_DEVICE_GRAPHS=2"; do
  echo -n "pcm [$setting]  "
  env $setting python3 bench.py --only pcm --steps 48 --warmup 8 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step'],3), round(d['value']), 'host enqueue ms/step', round(d['config']['host_enqueue_ms_per_step'],3))"
done
