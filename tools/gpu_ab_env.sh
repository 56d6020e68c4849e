#!/bin/bash
# legs of the bench under several environment settings: tools/gpu_ab_env.sh "LEGS" "A=1 B=2" "C=3" ...  ("-" = no setting)
LEGS=$1; shift
for rep in 1 2; do for leg in $LEGS; do for setting in "$@"; do
  echo -n "$leg [$setting]  "
  if [ "$setting" = "-" ]; then setting="VBM_NOTHING=1"; fi
  env $setting python3 bench.py --only $leg --steps 48 --warmup 8 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['ms_per_step'],3), round(d['value']))"
done; done; done
