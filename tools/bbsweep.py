"""Build time of one workload (frames one at a time) for the current environment. (GPU)"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, numpy as np
rt = importlib.import_module("ray-tracer_amd")
w = sys.argv[1]
scene, pkw, W, H, label = bench.build_workload(rt, w)
p = rt.default_params(W, H, **pkw)
out = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
for _ in range(24):
    scene.render_device(p, rebuild_grid=True, d_u8=out.data_ptr())
torch.cuda.synchronize()
st = [scene.stats_at(i) for i in range(20)]
print(w, "build %.4f render %.4f" % (np.mean([x["ms_build"] for x in st]), np.mean([x["ms_render"] for x in st])), flush=True)
