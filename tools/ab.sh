#!/bin/bash
# A/B of library variants: LIBS="a.so b.so" tools/ab.sh workload...
for l in $LIBS; do
  for w in "$@"; do
    RT_AMD_LIB=$PWD/ray-tracer_amd/$l timeout -k 10 300 python -u bench.py --no-cpu --workload $w --steps 20 2>gpurun_out/ab.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$l $w', d['value'], d['frame_ms'])" || { tail -5 gpurun_out/ab.err; exit 1; }
  done
done
