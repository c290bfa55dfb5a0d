"""Random views of the reference's scene against the oracle (GPU + CPU oracle): python tools/fuzz.py [n_views] [seed] [subdivisions]"""
import importlib, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import oracle_py as O
rt = importlib.import_module("ray-tracer_amd")
MESH = os.path.join(ROOT, "tests", "golden", "meshes")
SPOT, BLUB = os.path.join(MESH, "spot_triangulated.obj"), os.path.join(MESH, "blub_triangulated.obj")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
s = rt.Scene(); o = O.OracleScene()
if len(sys.argv) > 3:       # third argument k: the blub mesh subdivided k times (crowded voxels), beside spot
    tmp = rt.Scene(); tmp.load_mesh(BLUB); v9 = rt.subdivide(tmp.triangles(), int(sys.argv[3])); tmp.close()
    v9 = v9.copy(); v9[:, 0::3] += 1.5
    s.load_mesh(SPOT); s.add_triangles(v9, (30, 200, 90)); o.add_obj(SPOT); o.add_triangles(v9, (30, 200, 90))
else:
    s.load_mesh(SPOT); s.load_mesh(BLUB, True, (30, 200, 90), (1.5, 0, 0))
    o.add_obj(SPOT); o.add_obj(BLUB, True, (30, 200, 90), (1.5, 0, 0))
bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
bad = 0
for k in range(n):
    r = rng.uniform(0.3, 12.0); th = rng.uniform(0, 2 * np.pi); ph = rng.uniform(-1.2, 1.2)
    cam = (0.75 + r * np.cos(th) * np.cos(ph), r * np.sin(ph), r * np.sin(th) * np.cos(ph))
    kw = dict(cam_pos=tuple(float(x) for x in cam), cam_target=(float(rng.uniform(-0.5, 2.0)), float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.5, 0.5))),
              fov=float(rng.uniform(5, 110)), light_pos=tuple(float(x) for x in rng.uniform(-8, 8, 3)), eps=float(rng.choice([0.1, 0.01, 0.5])))
    W, H = int(rng.integers(17, 400)), int(rng.integers(9, 300))
    o_rgb, o_ids, _ = o.render(O.default_params(W, H, walk_mode=0, **kw))
    rgb, _, ids = s.render(rt.default_params(W, H, walk_mode=rt.RT_WALK_FULL, **kw))
    d_full = int((ids != o_ids).sum()); c_full = int((bits(rgb) != bits(o_rgb)).any(axis=-1).sum())
    rgb, _, ids = s.render(rt.default_params(W, H, walk_mode=rt.RT_WALK_EARLY, **kw))
    d_early = int((ids != o_ids).sum())
    flag = "" if (d_full == 0 and c_full == 0 and d_early <= 3) else "   <-- MISMATCH"
    bad += bool(flag)
    print("view %2d %dx%d hits %6d | full walk: ids differ %d, colour bits differ %d | early exit: ids differ %d%s" % (k, W, H, int((o_ids >= 0).sum()), d_full, c_full, d_early, flag), flush=True)
print("views with mismatches:", bad)
sys.exit(1 if bad else 0)
