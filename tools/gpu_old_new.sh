#!/bin/bash
# the same leg from several trees on one box: tools/gpu_old_new.sh LEG dir1 dir2 ...  (built worktrees inside the repo)
LEG=$1; shift
run() { (cd $1 && VBM_NOISE_RING=0 python3 bench.py --only $2 --steps 48 --warmup 8 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['ms_per_step'],3), round(d['value']))"); }
for rep in 1 2 3; do for d in "$@"; do
  echo -n "$d $LEG: "; run $d $LEG
done; done
