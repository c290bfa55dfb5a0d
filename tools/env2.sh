#!/bin/bash
# bench one workload under several environment settings: tools/env2.sh workload "A=1 B=2" "A=3" ...
mkdir -p gpurun_out
w=$1; shift
for cfg in "$@"; do
  env $cfg timeout -k 10 200 python -u bench.py --no-cpu --workload $w --steps 20 2>gpurun_out/env.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg $w', d['value'], d['frame_ms'])" || { tail -5 gpurun_out/env.err; exit 1; }
done
