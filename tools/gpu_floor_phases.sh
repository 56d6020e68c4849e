#!/bin/bash
# floor fit (cooperative kernel) with phases switched off: where does its time go?  (results are wrong with phases off)
for p in 31 0 1 3 7 11 15 19 27; do
  echo -n "phases=$p "; VBM_FLOORFIT_PHASES=$p VBM_BENCH_SOLO_STEPS=8 python3 bench.py --only solo 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['stage_solo_ms']['floor_fit'],4))"
done
