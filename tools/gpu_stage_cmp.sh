#!/bin/bash
# stage times (HIP events in situ, per-block leg) from several trees: tools/gpu_stage_cmp.sh dir1 dir2 ...
for d in "$@"; do
  (cd $d && VBM_NOISE_RING=0 python3 bench.py --only block --steps 48 --warmup 8 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$d', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['stage_ms_per_launch'].items()})")
  (cd $d && VBM_NOISE_RING=0 python3 bench.py --only solo --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$d solo', {k: round(v,3) for k,v in d['stage_solo_ms'].items()})")
done
