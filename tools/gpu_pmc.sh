#!/bin/bash
# usage: tools/gpu_pmc.sh "<counters pass 1>" ["<counters pass 2>" ...] — one rocprofv3 --pmc pass per argument on a short bench
cd /tmp && export TMPDIR=/tmp
i=0
for ctrs in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --pmc $ctrs --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$i -o p -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_$i.log 2>&1
  echo "pass $i rc=$?"
done
ls -la $GRAFT_REPO_ROOT/gpurun_out/pmc_*/
