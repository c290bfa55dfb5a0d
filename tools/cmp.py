"""Render one workload under two settings of an environment variable and compare hit ids and colours (GPU):
   python tools/cmp.py workload VAR value_a value_b"""
import importlib, sys, os, subprocess, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 2 and sys.argv[1] == "--dump":
    import bench
    rt = importlib.import_module("ray-tracer_amd")
    scene, pkw, W, H, label = bench.build_workload(rt, sys.argv[2])
    p = rt.default_params(W, H, **pkw)
    rgb, u8, ids = scene.render(p, want_u8=True)
    st = scene.last_stats()
    print("hits", st["hits"], "shadowed", st["shadowed"])
    np.save(sys.argv[3] + "_ids.npy", ids); np.save(sys.argv[3] + "_rgb.npy", rgb)
    sys.exit(0)
w, var, va, vb = sys.argv[1:5]
for tag, v in (("a", va), ("b", vb)):
    subprocess.check_call([sys.executable, __file__, "--dump", w, "gpurun_out/cmp_" + tag], env=dict(os.environ, **{var: v}))
a = np.load("gpurun_out/cmp_a_ids.npy"); b = np.load("gpurun_out/cmp_b_ids.npy")
ra = np.load("gpurun_out/cmp_a_rgb.npy"); rb = np.load("gpurun_out/cmp_b_rgb.npy")
d = np.argwhere(a != b)
print("differing ids", len(d), "of", a.size)
for y, x in d[:12]: print("  y", y, "x", x, var, va, "->", a[y, x], "|", vb, "->", b[y, x], "tile", (int(y) // 8, int(x) // 8), "in-tile", (int(y) % 8, int(x) % 8), "rgb", ra[y, x].tolist(), rb[y, x].tolist())
c = np.argwhere((ra.view(np.uint32) != rb.view(np.uint32)).any(axis=-1) & (a == b))
print("same id, other colour:", len(c))
for y, x in c[:12]: print("  y", y, "x", x, "id", a[y, x], ra[y, x], rb[y, x])
