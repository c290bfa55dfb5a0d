#!/bin/bash
# quick GPU check: parity tests, then compact bench numbers for the listed workloads
mkdir -p gpurun_out
timeout -k 10 400 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1
tail -4 gpurun_out/pytest_gpu.log
for w in "$@"; do
  timeout -k 10 300 python -u bench.py --no-cpu --workload $w --steps 20 2>gpurun_out/qb_$w.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$w', d['value'], d['frame_ms'], 'frac', d['roofline']['frac'])" || tail -5 gpurun_out/qb_$w.err
done
