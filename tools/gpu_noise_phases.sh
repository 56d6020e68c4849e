#!/bin/bash
# solo time of the noise mask kernel with phases switched off (VBM_NOISE_PHASES bits: 1 scan, 2 solve, 4 M1/companding rows, 8 M2/M8, 16 rows out)
for ring in 0 1; do for v in 31 30 29 27 23 15 28 0; do
  echo -n "ring=$ring phases=$v  "
  VBM_NOISE_RING=$ring VBM_NOISE_PHASES=$v python3 bench.py --only solo --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['stage_solo_ms']['noisemask'],3))"
done; done
