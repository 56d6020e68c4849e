#!/bin/bash
# two bench processes side by side on one GPU, half the streams each: does more independent work in flight raise the aggregate?
run() { VBM_BENCH_STREAMS=$1 python3 bench.py --only pcm --steps ${2:-96} --warmup 8 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('streams', d['config']['streams_per_gpu'], 'ms', round(d['ms_per_step'],3), 'value', round(d['value']), 'enc/in', round(d['config']['encoded_over_input'],4))"; }
echo "one process, 16384:"; run 16384
echo "one process, 8192:"; run 8192
echo "two processes, 8192 each:"; run 8192 200 & run 8192 200 & wait
echo "three processes, 5440 each:"; run 5440 200 & run 5440 200 & run 5440 200 & wait
