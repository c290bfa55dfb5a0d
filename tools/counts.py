"""Counters of one counting frame of the default workload (steps, tests, trips, phase cycles): python tools/counts.py (GPU)"""
import importlib, sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
rt = importlib.import_module("ray-tracer_amd")
scene, pkw, W, H, label = bench.build_workload(rt, "nefertiti4k")
scene.upload()
p = rt.default_params(W, H, **pkw)
buf = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
scene.render_device(p, rebuild_grid=True, d_u8=buf.data_ptr(), stream=st)
scene.set_counting(True)
scene.render_device(p, rebuild_grid=False, d_u8=buf.data_ptr(), stream=st)
torch.cuda.synchronize()
cs = scene.last_stats()
for k in sorted(cs): print(k, cs[k])
