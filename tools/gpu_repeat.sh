#!/bin/bash
# run-to-run spread of a leg: tools/gpu_repeat.sh LEG N [ENV=VALUE ...]
LEG=$1; N=$2; shift 2
for i in $(seq $N); do
  env "$@" python3 bench.py --only $LEG --steps 48 --warmup 8 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), end=' ')"
done; echo
