#!/bin/bash
# bench workloads with several prebuilt library variants: tools/kvar.sh "lib1.so lib2.so" workload...   (RT_AMD_LIB selects the library)
mkdir -p gpurun_out
libs=$1; shift
for l in $libs; do
  for w in "$@"; do
    RT_AMD_LIB=$l timeout -k 10 200 python -u bench.py --no-cpu --workload $w --steps 20 2>gpurun_out/kvar.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$l $w', d['value'], d['frame_ms'])" || { tail -5 gpurun_out/kvar.err; exit 1; }
  done
done
