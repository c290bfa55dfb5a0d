"""rt_render_device_noalloc under HIP graph capture: capture one frame, replay it, compare with the plain call and time both. (GPU)"""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, numpy as np
rt = importlib.import_module("ray-tracer_amd")
w = sys.argv[1] if len(sys.argv) > 1 else "spot1080"
scene, pkw, W, H, label = bench.build_workload(rt, w)
p = rt.default_params(W, H, **pkw)
scene.upload()
out = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
ref = torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda")
scene.render_device(p, rebuild_grid=True, d_u8=ref.data_ptr())
torch.cuda.synchronize()
st = torch.cuda.Stream()
g = torch.cuda.CUDAGraph()
with torch.cuda.stream(st):
    scene.render_device_noalloc(p, d_u8=out.data_ptr(), stream=st.cuda_stream)      # warm-up outside the capture
    st.synchronize()
    g.capture_begin()
    scene.render_device_noalloc(p, d_u8=out.data_ptr(), stream=st.cuda_stream)
    g.capture_end()
out.zero_(); torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("replayed frame equals the plain frame:", bool(torch.equal(out, ref)))
def timeit(f, n=200):
    for _ in range(10): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("graph replay   %.4f ms/frame" % timeit(g.replay))
print("noalloc direct %.4f ms/frame" % timeit(lambda: scene.render_device_noalloc(p, d_u8=out.data_ptr(), stream=st.cuda_stream)))
print("render_device  %.4f ms/frame" % timeit(lambda: scene.render_device(p, rebuild_grid=True, d_u8=out.data_ptr(), stream=st.cuda_stream)))
