#!/bin/bash
# kernel iteration check: the GPU parity suite, then bench lines (no CPU leg) and the counting frame's loop counts
mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -3 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
for w in "$@"; do
  timeout -k 10 300 python -u bench.py --no-cpu --no-extras --workload $w --steps 40 2>gpurun_out/qk_$w.err | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); o=d['frame_ms']['one_at_a_time']
print('$w', 'in-flight', d['value'], 'Mrays/s', d['ms_per_step'], 'ms | one at a time', o['ms_per_frame'], 'build', o['build'], 'render', o['render'], '| frac', d['roofline']['frac'])" || { tail -5 gpurun_out/qk_$w.err; exit 1; }
done
timeout -k 10 200 python tools/counts.py nefertiti4k > gpurun_out/counts_nef.txt 2>&1 && grep "loop\|ms_render" gpurun_out/counts_nef.txt
