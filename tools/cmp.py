"""Render one workload twice in separate processes' settings (env var given on the command line) and compare ids (GPU)."""
import importlib, sys, os, subprocess, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--dump":
    import bench
    rt = importlib.import_module("ray-tracer_amd")
    scene, pkw, W, H, label = bench.build_workload(rt, sys.argv[2])
    p = rt.default_params(W, H, **pkw)
    rgb, u8, ids = scene.render(p, want_u8=True)
    st = scene.last_stats()
    print("continued rays", st.get("continued_rays"), "hits", st["hits"])
    np.save(sys.argv[3], ids)
    sys.exit(0)
w = sys.argv[1]
env0 = dict(os.environ, RT_SPLIT_CYCLES="0"); env1 = dict(os.environ)
subprocess.check_call([sys.executable, __file__, "--dump", w, "gpurun_out/ids0.npy"], env=env0)
subprocess.check_call([sys.executable, __file__, "--dump", w, "gpurun_out/ids1.npy"], env=env1)
a = np.load("gpurun_out/ids0.npy"); b = np.load("gpurun_out/ids1.npy")
d = np.argwhere(a != b)
print("differing pixels", len(d), "of", a.size)
for y, x in d[:12]: print("  y", y, "x", x, "ref", a[y, x], "got", b[y, x], "tile", (y // 8, x // 8), "in-tile", (y % 8, x % 8))
