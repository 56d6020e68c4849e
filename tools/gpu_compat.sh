#!/bin/bash
# the drop-in path: threads x streams, both delivery modes, with the host time breakdown
export LD_LIBRARY_PATH=$PWD/vorbis_aotuv_lancer_amd:$LD_LIBRARY_PATH
for defer in ${DEFER:-0 1}; do
  echo "== VORBIS_MI355X_DEFER_BLOCKS=$defer: ${ARGS:-4 4096 4096 24 8}"
  VORBIS_MI355X_DEFER_BLOCKS=$defer timeout -k 10 300 examples/compat_bench ${ARGS:-4 4096 4096 24 8}
done
