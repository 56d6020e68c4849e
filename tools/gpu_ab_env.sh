#!/bin/bash
# A/B of an environment switch on both timed legs: tools/gpu_ab_env.sh VAR=VALUE [steps]
SW=$1; STEPS=${2:-48}
for leg in block pcm; do
  for v in "" "$SW"; do
    echo -n "$leg [${v:-default}] "
    env $v python3 bench.py --only $leg --steps $STEPS --warmup 8 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step'],3), round(d['value']), d['config'].get('short_block_fraction'), d['config'].get('encoded_over_input'))"
  done
done
