#!/bin/bash
# sweep of the trace scheduling parameters over the listed workloads; each config is "thr0:thr1/shifts16hex/w0:w1:w2:w3"
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1 || { tail -20 gpurun_out/pytest_gpu.log; exit 1; }
tail -2 gpurun_out/pytest_gpu.log
for cfg in $SCHEDS; do
  IFS=/ read a b c <<< "$cfg"
  for w in "$@"; do
    RT_SCHED=$a RT_SHIFTS=$b RT_EST_W=$c timeout -k 10 300 python -u bench.py --no-cpu --workload $w --steps 20 2>gpurun_out/qb_$w.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$cfg $w', d['value'], d['frame_ms'])" || { tail -5 gpurun_out/qb_$w.err; exit 1; }
  done
done
