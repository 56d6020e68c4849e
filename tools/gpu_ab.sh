#!/bin/bash
# A/B of environment knobs on the whole bench: usage: gpu_ab.sh "ENV1=a ENV2=b" "ENV1=c" ...
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  env $cfg python3 bench.py --no-cpu-baseline ${BENCH_ARGS} 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
pb=d.get('per_block_path',{})
print('$cfg', '| pcm ms', round(d['ms_per_step'],3), 'value', round(d['value']), '| block ms', round(pb.get('ms_per_step',0),3), '| tone', round(d['stage_ms_per_step']['tonemask'],3), round(pb.get('stage_ms_per_step',{}).get('tonemask',0),3))"
done
