"""Random views, several frames in flight and random shares of the frame (indexed voxel lists), against the oracle:
python tools/fuzz_flight.py [n_views] [seed] [subdivisions]   (GPU + CPU oracle)"""
import importlib, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
import oracle_py as O
rt = importlib.import_module("ray-tracer_amd")
MESH = os.path.join(ROOT, "tests", "golden", "meshes")
SPOT, BLUB = os.path.join(MESH, "spot_triangulated.obj"), os.path.join(MESH, "blub_triangulated.obj")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
s = rt.Scene(); o = O.OracleScene()
if len(sys.argv) > 3:
    tmp = rt.Scene(); tmp.load_mesh(BLUB); v9 = rt.subdivide(tmp.triangles(), int(sys.argv[3])); tmp.close()
    v9 = v9.copy(); v9[:, 0::3] += 1.5
    s.load_mesh(SPOT); s.add_triangles(v9, (30, 200, 90)); o.add_obj(SPOT); o.add_triangles(v9, (30, 200, 90))
else:
    s.load_mesh(SPOT); s.load_mesh(BLUB, True, (30, 200, 90), (1.5, 0, 0))
    o.add_obj(SPOT); o.add_obj(BLUB, True, (30, 200, 90), (1.5, 0, 0))
s.upload()
depth = 4
s.set_frames_in_flight(depth)
lanes = rt.create_streams(depth)
bits = lambda a: np.ascontiguousarray(a, np.float32).view(np.uint32)
views = []
for k in range(n):
    r = rng.uniform(0.3, 12.0); th = rng.uniform(0, 2 * np.pi); ph = rng.uniform(-1.2, 1.2)
    cam = (0.75 + r * np.cos(th) * np.cos(ph), r * np.sin(ph), r * np.sin(th) * np.cos(ph))
    kw = dict(cam_pos=tuple(float(x) for x in cam), cam_target=(float(rng.uniform(-0.5, 2.0)), float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-0.5, 0.5))),
              fov=float(rng.uniform(5, 110)), light_pos=tuple(float(x) for x in rng.uniform(-8, 8, 3)), eps=float(rng.choice([0.1, 0.01, 0.5])))
    W, H = int(rng.integers(17, 400)), int(rng.integers(9, 300))
    parts = int(rng.choice([1, 1, 2, 3, 8])); part = int(rng.integers(0, parts))
    o_rgb, o_ids, _ = o.render(O.default_params(W, H, walk_mode=0, **kw))
    p = rt.default_params(W, H, walk_mode=rt.RT_WALK_FULL, part_index=part, part_count=parts, **kw)
    rows = rt.part_rows(p)
    views.append((p, o_rgb, o_ids, torch.full((rows, W, 3), -1.0, dtype=torch.float32, device="cuda"), torch.full((rows, W), -9, dtype=torch.int32, device="cuda")))
torch.cuda.synchronize()
for rep in range(2):                      # twice: the second time every working set has a grid of another view's mode in it
    for k, (p, _, _, rgb_d, ids_d) in enumerate(views):
        s.render_device(p, rebuild_grid=bool(rng.integers(0, 2)), d_rgb=rgb_d.data_ptr(), d_ids=ids_d.data_ptr(), stream=lanes[k % len(lanes)])
torch.cuda.synchronize()
bad = 0
for k, (p, o_rgb, o_ids, rgb_d, ids_d) in enumerate(views):
    m = rt.part_row_map(p)
    rgb, ids = rgb_d.cpu().numpy()[m >= 0], ids_d.cpu().numpy()[m >= 0]
    w_rgb, w_ids = o_rgb[m[m >= 0]], o_ids[m[m >= 0]]
    d_id = int((ids != w_ids).sum()); d_c = int((bits(rgb) != bits(w_rgb)).any(axis=-1).sum())
    flag = "" if (d_id == 0 and d_c == 0) else "   <-- MISMATCH"
    bad += bool(flag)
    print("view %2d %dx%d part %d/%d hits %6d | ids differ %d, colour bits differ %d%s" % (k, p.width, p.height, p.part_index, p.part_count, int((w_ids >= 0).sum()), d_id, d_c, flag), flush=True)
print("views with mismatches:", bad)
rt.destroy_streams(lanes)
sys.exit(1 if bad else 0)
