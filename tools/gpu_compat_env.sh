#!/bin/bash
# the drop-in path under environment settings: tools/gpu_compat_env.sh "A=1 B=2" "C=3" ...  ("-" = none)
export LD_LIBRARY_PATH=$PWD/vorbis_aotuv_lancer_amd:$LD_LIBRARY_PATH
for setting in "$@"; do
  echo -n "[$setting] "
  if [ "$setting" = "-" ]; then setting="VBM_NOTHING=1"; fi
  env VORBIS_MI355X_DEFER_BLOCKS=${DEFER:-1} $setting timeout -k 10 300 examples/compat_bench ${ARGS:-4 4096 4096 24 8} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), 'streams at 1x;', d['device_rounds_total'], 'rounds; wall', d['wall_s'])"
done
