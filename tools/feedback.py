"""What the render kernel reports about its triangle tests per workload (record runs streamed, trips with a candidate
triangle) and whether the next frame takes the kernel variant that finishes hit trips together (python tools/feedback.py
workload...) (GPU)"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
rt = importlib.import_module("ray-tracer_amd")
for w in sys.argv[1:]:
    scene, pkw, W, H, label = bench.build_workload(rt, w)
    p = rt.default_params(W, H, **pkw)
    out = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
    for _ in range(3):
        scene.render_device(p, rebuild_grid=True, d_u8=out.data_ptr()); torch.cuda.synchronize()
    r, h, d = scene.test_feedback()
    print(w, "runs", r, "hit trips", h, "per run %.2f" % (h / max(r, 1)), "deferred next:", d, flush=True)
