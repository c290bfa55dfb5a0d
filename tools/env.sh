#!/bin/bash
# bench the listed workloads under several values of one environment variable: VAR=NAME VALS="a b c" tools/env.sh workload...
mkdir -p gpurun_out
for v in $VALS; do
  for w in "$@"; do
    env $VAR=$v timeout -k 10 200 python -u bench.py --no-cpu --workload $w --steps 20 2>gpurun_out/env.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$VAR=$v $w', d['value'], d['frame_ms'])" || { tail -5 gpurun_out/env.err; exit 1; }
  done
done
