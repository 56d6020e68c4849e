#!/bin/bash
# noise mask alone at other occupancies: LDS padding (fewer workgroups per CU) and one block per workgroup (more)
run() { echo -n "$* : "; env "$@" python3 bench.py --only solo --steps 12 --warmup 4 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print(round(d['stage_solo_ms']['noisemask'],3))"; }
run VBM_NOISE_LDS_PAD=0
run VBM_NOISE_LDS_PAD=16
run VBM_NOISE_LDS_PAD=40
run VBM_NOISE_NB1=1
run VBM_NOISE_NB1=1 VBM_NOISE_LDS_PAD=8
