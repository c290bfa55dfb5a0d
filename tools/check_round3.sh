mkdir -p gpurun_out
timeout -k 10 500 python -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > gpurun_out/pytest_gpu.log 2>&1; rc=$?
tail -5 gpurun_out/pytest_gpu.log
[ $rc -eq 0 ] || exit 1
RT_BENCH_BACKEND=gloo timeout -k 10 250 python3 bench.py --gpus 2 --steps 5 --no-cpu > gpurun_out/selflaunch2.json 2> gpurun_out/selflaunch2.err || { tail -20 gpurun_out/selflaunch2.err; exit 1; }
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/selflaunch2.json') if l.startswith('{')][-1]); print('self-launched ranks', d['n_gpus'], 'frame_check', d['frame_check'], 'ms', d['ms_per_step'])"
timeout -k 10 280 python3 bench.py --steps 20 > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err || { tail -20 gpurun_out/bench_default.err; exit 1; }
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/bench_default.json') if l.startswith('{')][-1])
for k in ('metric','value','ms_per_step','value_host_visible','ms_per_step_host_visible','frame_ms_latency','full_walk'): print(k, d[k])
print('roofline', {k: d['roofline'][k] for k in ('frac','launch_ms','gpu_throughput_frac','traffic')}); print('parity', d['parity']); print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['kind'])"
