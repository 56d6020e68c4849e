#!/bin/bash
# usage: tools/gpu_sweep.sh  — parity test, then bench with a few lanes-per-wave settings (stage times only)
set -o pipefail
cd "$GRAFT_REPO_ROOT"
timeout -k 10 600 python -m pytest tests/test_pipeline_gpu.py -x -q > gpurun_out/t1.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/t1.log
for sp in 1; do
  VBM_OVERLAP_BRANCHES=0 VBM_SUB_BATCHES=$sp timeout -k 10 200 python bench.py --steps 16 --warmup 3 --no-cpu-baseline > gpurun_out/spn_$sp.log 2>&1; grep -o "ms_per_step[^,]*" gpurun_out/spn_$sp.log; VBM_SUB_BATCHES=$sp timeout -k 10 200 python bench.py --steps 16 --warmup 3 --no-cpu-baseline > gpurun_out/sp_$sp.log 2>&1
  python3 - "$sp" <<'PY'
import json,sys
a=sys.argv[1]
for line in open(f"gpurun_out/sp_{a}.log"):
    if line.startswith("{"):
        d=json.loads(line)
        print(f"split={a} ms/step={d['ms_per_step']:.2f} value={d['value']:.0f}")
PY
done
