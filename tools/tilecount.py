"""Listed tiles of the bench workloads (counting build of the trace kernel). (GPU)"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
rt = importlib.import_module("ray-tracer_amd")
for w in sys.argv[1:]:
    scene, pkw, W, H, label = bench.build_workload(rt, w)
    for parts in (1, 8):
        p = rt.default_params(W, H, part_index=0, part_count=parts, **pkw)
        out = torch.empty((rt.part_rows(p), W, 3), dtype=torch.uint8, device="cuda")
        scene.set_counting(True)
        scene.render_device(p, rebuild_grid=True, d_u8=out.data_ptr())
        torch.cuda.synchronize()
        c = scene.last_stats()
        print(w, "parts", parts, "tiles traced", c["tiles_traced"], "of", ((W + 7) // 8) * ((rt.part_rows(p) + 7) // 8), "refs", c["n_refs"], "cells", c["n_cells"], flush=True)
        scene.set_counting(False)
