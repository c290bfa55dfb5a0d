"""Build, trace and total time of every frame of one burst of 20 stand-in frames, four in flight (python tools/burstlat.py) (GPU)"""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
rt = importlib.import_module("ray-tracer_amd")
scene, pkw, W, H, label = bench.build_workload(rt, "nefertiti4k")
p = rt.default_params(W, H, **pkw)
scene.upload(); scene.set_frames_in_flight(4)
hs = rt.create_streams(4)
outs = [torch.empty((H, W, 3), dtype=torch.uint8, device="cuda") for _ in hs]
def run(m):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(m):
        scene.render_device(p, rebuild_grid=True, d_u8=outs[i % 4].data_ptr(), stream=hs[i % 4])
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
run(40); run(40)
t = run(20)
st = [scene.stats_at(i) for i in range(20)][::-1]
print("burst of 20: %.3f ms total" % t)
for i, x in enumerate(st):
    print(i, "build %.3f trace %.3f total %.3f" % (x["ms_build"], x["ms_render"], x["ms_total"]))
