#!/bin/bash
# one leg of the bench under several values of an environment switch: tools/gpu_ab_vals.sh LEG VAR v1 v2 ... (repeats each twice)
LEG=$1; VAR=$2; shift 2
for rep in 1 2; do for v in "$@"; do
  echo -n "$LEG $VAR=$v  "
  env $VAR=$v python3 bench.py --only $LEG --steps 48 --warmup 8 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step'],3), round(d['value']), d['config'].get('short_block_fraction'), d['config'].get('encoded_over_input'))"
done; done
