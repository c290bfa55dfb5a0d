#!/bin/bash
# N > 1 code path of bench.py on a box with one GPU: ranks share the GPU, the gather goes over gloo (numbers mean nothing)
mkdir -p gpurun_out
for n in "$@"; do
  RT_BENCH_BACKEND=gloo timeout -k 10 280 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) bench.py --gpus $n --steps 6 --warmup 3 --no-cpu > gpurun_out/rehearsal_$n.json 2> gpurun_out/rehearsal_$n.err || { tail -20 gpurun_out/rehearsal_$n.err; exit 1; }
  python3 -c "
import json,sys
d=json.loads([l for l in open('gpurun_out/rehearsal_$n.json') if l.startswith('{')][-1]); print('ranks', d['n_gpus'], 'frame_check', d['frame_check'], 'ms', d['ms_per_step'], d['frame_ms'])"
done
# the same path through RCCL itself, one rank (process group on the nccl backend, gather on the lanes' streams, frame check)
RT_BENCH_FORCE_EXCHANGE=1 timeout -k 10 280 python bench.py --no-cpu --no-extras --steps 12 > gpurun_out/rehearsal_rccl1.json 2> gpurun_out/rehearsal_rccl1.err || { tail -20 gpurun_out/rehearsal_rccl1.err; exit 1; }
python3 -c "
import json
d=json.loads([l for l in open('gpurun_out/rehearsal_rccl1.json') if l.startswith('{')][-1]); print('rccl, 1 rank: frame_check', d['frame_check'], 'ms', d['ms_per_step'])"
