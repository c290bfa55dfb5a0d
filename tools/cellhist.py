"""Histogram of the voxel list lengths of a workload's grid. (GPU)"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
rt = importlib.import_module("ray-tracer_amd")
for w in sys.argv[1:]:
    scene, pkw, W, H, label = bench.build_workload(rt, w)
    scene.build_grid()
    start, tris = scene.grid_csr()
    n = np.diff(start.astype(np.int64)); n = n[n > 0]
    print(w, "occupied voxels", len(n), "mean", n.mean(), "median", np.median(n))
    for lim in (8, 16, 24, 32, 48, 63, 64, 128, 256, 1 << 30):
        print("   <= %-10d voxels %.3f  refs %.3f" % (lim, (n <= lim).mean(), n[n <= lim].sum() / n.sum()))
