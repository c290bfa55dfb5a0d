"""Experiment: two scenes of the same mesh rendered alternately on two streams -- how much does overlapping one frame's
grid build with the other frame's trace tail buy? (GPU)"""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
rt = importlib.import_module("ray-tracer_amd")
w = sys.argv[1] if len(sys.argv) > 1 else "nefertiti4k"
parts = int(sys.argv[2]) if len(sys.argv) > 2 else 1          # > 1: one rank's share of the frame (part 0 of N)
n_scenes = int(sys.argv[3]) if len(sys.argv) > 3 else 2
scenes = []
for k in range(n_scenes):
    scene, pkw, W, H, label = bench.build_workload(rt, w)
    p = rt.default_params(W, H, part_index=0, part_count=parts, **pkw)
    out = torch.empty((rt.part_rows(p), W, 3), dtype=torch.uint8, device="cuda")
    st = torch.cuda.Stream()
    scene.upload()
    scenes.append((scene, p, out, st))
def run(n, two):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(n):
        scene, p, out, st = scenes[i % n_scenes] if two else scenes[0]
        scene.render_device(p, rebuild_grid=True, d_u8=out.data_ptr(), stream=st.cuda_stream)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
run(6, True)
print(w, "parts", parts, "scenes", n_scenes, "one scene, one stream: %.4f ms/frame | two scenes alternating on two streams: %.4f ms/frame" % (run(40, False), run(40, True)))
