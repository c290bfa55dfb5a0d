"""Build and trace time of the stand-in with two scene-spanning triangles (a ground quad) added: the thread-per-triangle build
kernels walk a triangle's whole voxel range in one lane. python tools/bigtri.py [subdivisions] (GPU)"""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
rt = importlib.import_module("ray-tracer_amd")
k = int(sys.argv[1]) if len(sys.argv) > 1 else 4
tmp = rt.Scene(); tmp.load_mesh(os.path.join(bench.MESH, "blub_triangulated.obj"))
v9 = rt.subdivide(tmp.triangles(), k); tmp.close()
lo, hi = v9.reshape(-1, 3).min(0), v9.reshape(-1, 3).max(0)
y = hi[1] + 0.01          # the Serial camera looks with up = (0,-1,0): "below" the mesh is +y
a, b, c, d = (lo[0] - 1, y, lo[2] - 1), (hi[0] + 1, y, lo[2] - 1), (hi[0] + 1, y, hi[2] + 1), (lo[0] - 1, y, hi[2] + 1)
quad = np.array([a + b + c, a + c + d], dtype=np.float32)
W, H = 3840, 2160
out = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
for name, arr in (("mesh alone", v9), ("mesh + ground quad", np.concatenate([v9, quad]))):
    s = rt.Scene(); s.add_triangles(np.ascontiguousarray(arr)); s.upload()
    p = rt.default_params(W, H)
    for _ in range(8):
        s.render_device(p, rebuild_grid=True, d_u8=out.data_ptr())
    torch.cuda.synchronize()
    st = [s.stats_at(i) for i in range(5)]
    print(name, "build %.3f trace %.3f ms" % (np.mean([x["ms_build"] for x in st]), np.mean([x["ms_render"] for x in st])), "refs", st[0]["n_refs"], "hits", st[0]["hits"], flush=True)
    s.close()
