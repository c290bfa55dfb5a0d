"""Experiment: which streams overlap? n scenes of one mesh rendered round-robin, each on its own stream; the streams are taken
from a list created up front (python tools/overlap2.py workload parts n_scenes stream-indices...)."""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
rt = importlib.import_module("ray-tracer_amd")
w, parts, n = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
idx = [int(x) for x in sys.argv[4:]] or list(range(n))
pool = [torch.cuda.Stream() for _ in range(max(idx) + 1)]
scenes = []
for k in range(n):
    scene, pkw, W, H, label = bench.build_workload(rt, w)
    p = rt.default_params(W, H, part_index=0, part_count=parts, **pkw)
    out = torch.empty((rt.part_rows(p), W, 3), dtype=torch.uint8, device="cuda")
    scene.upload()
    scenes.append((scene, p, out, pool[idx[k]]))
def run(m):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(m):
        scene, p, out, st = scenes[i % n]
        scene.render_device(p, rebuild_grid=True, d_u8=out.data_ptr(), stream=st.cuda_stream)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / m * 1e3
run(2 * n)
print(w, "parts", parts, "scenes", n, "streams", idx, "%.4f ms/frame" % run(60), flush=True)
