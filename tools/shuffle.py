"""Frame time of the stand-in with its triangles in file order and in random order (how much the run culling of crowded voxels
depends on the coherence of the file order): python tools/shuffle.py [subdivisions] (GPU)"""
import importlib, sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
rt = importlib.import_module("ray-tracer_amd")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
tmp = rt.Scene(); tmp.load_mesh(os.path.join(bench.MESH, "blub_triangulated.obj"))
v9 = rt.subdivide(tmp.triangles(), k); tmp.close()
W, H = 3840, 2160
out = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
ref = None
for name, arr in (("file order", v9), ("random order", v9[np.random.default_rng(1).permutation(v9.shape[0])])):
    s = rt.Scene(); s.add_triangles(np.ascontiguousarray(arr)); s.upload()
    p = rt.default_params(W, H)
    for _ in range(12):
        s.render_device(p, rebuild_grid=True, d_u8=out.data_ptr())
    torch.cuda.synchronize()
    st = [s.stats_at(i) for i in range(8)]
    img = out.cpu().numpy().copy()
    if ref is None: ref = img
    s.set_counting(True); s.render_device(p, rebuild_grid=False, d_u8=out.data_ptr()); torch.cuda.synchronize()
    cs = s.last_stats(); s.set_counting(False)
    print(name, "build %.3f trace %.3f ms" % (np.mean([x["ms_build"] for x in st]), np.mean([x["ms_render"] for x in st])),
          "tests executed", cs["tests_executed"], "trips", cs["run_ray_trips"], "runs", cs["runs_streamed"], "record order", s.record_order(), "triangle order", s.triangle_order(),
          "image equal to file order:", bool(np.array_equal(img, ref)), flush=True)
    s.close()
