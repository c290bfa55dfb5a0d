"""Frame time of ONE rank's share of a frame (part 0 of N bands partitions), on one GPU: python tools/partbench.py workload N..."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
rt = importlib.import_module("ray-tracer_amd")
w = sys.argv[1]
scene, pkw, W, H, label = bench.build_workload(rt, w)
scene.upload()
print(w, 'record order (area unsorted, sorted, sort?):', scene.record_order(), flush=True)
stream = torch.cuda.current_stream().cuda_stream
for n in [int(x) for x in sys.argv[2:]]:
    for part in (0, n - 1):
        p = rt.default_params(W, H, part_index=part, part_count=n, **pkw)
        rows = rt.part_rows(p)
        buf = torch.empty((rows, W, 3), dtype=torch.uint8, device="cuda")
        for _ in range(23):
            scene.render_device(p, rebuild_grid=True, d_u8=buf.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        st = [scene.stats_at(i) for i in range(20)]
        print(w, "part %d of %d:" % (part, n), "build %.3f trace %.3f ms" % (np.mean([x["ms_build"] for x in st]), np.mean([x["ms_render"] for x in st])), flush=True)
