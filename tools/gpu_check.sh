#!/bin/bash
# solo stages, then the two legs twice
python3 bench.py --only solo --steps 16 --warmup 4 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k: round(v,3) for k,v in d['stage_solo_ms'].items()})"
bash tools/gpu_ab_vals.sh pcm ${VAR:-VBM_NOISE_SWZ} ${VALS:-0 1}
bash tools/gpu_ab_vals.sh block ${VAR:-VBM_NOISE_SWZ} ${VALS:-0 1}
