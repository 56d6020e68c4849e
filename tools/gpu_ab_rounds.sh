#!/bin/bash
for pat in "$@"; do
  echo -n "rounds=$pat  "
  VBM_BENCH_ROUNDS=$pat python3 bench.py --only pcm --steps ${STEPS:-96} --warmup 8 --no-cpu-baseline 2>&1 | tail -1 | python3 -c "
import json,sys
try:
    d=json.loads(sys.stdin.read())
    print(round(d['ms_per_step'],3), round(d['value']), round(d['config'].get('short_block_fraction'),3), round(d['config'].get('encoded_over_input'),4), d['config'].get('max_buffered_samples_at_end'))
except Exception as e: print('failed', e)"
done
