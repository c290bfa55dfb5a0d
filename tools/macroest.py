"""Estimate of what a coarse, dilated occupancy test per ray could cull: for a sample of the primary rays of a workload, the
steps their walk takes through the grid, whether it meets an occupied voxel at all, and whether a walk through macro cells
of m voxels (occupancy dilated by one voxel) meets an occupied one. numpy model of the walk (exact enough for an estimate).
python tools/macroest.py workload [stride]   (GPU for the grid only)"""
import importlib, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, bench
rt = importlib.import_module("ray-tracer_amd")
w = sys.argv[1]; stride = int(sys.argv[2]) if len(sys.argv) > 2 else 4
scene, pkw, W, H, label = bench.build_workload(rt, w)
scene.build_grid()
gi = scene.grid_info(); start, _ = scene.grid_csr()
n = gi["n"].astype(np.int64); lo = gi["lohi"][:3].astype(np.float64); hi = gi["lohi"][3:].astype(np.float64); wd = gi["width"].astype(np.float64)
occ = (np.diff(start.astype(np.int64)) > 0).reshape(n[2], n[1], n[0])          # z, y, x
def dilate(a):
    p = np.pad(a, 1)
    out = np.zeros_like(a)
    for dz in range(3):
        for dy in range(3):
            for dx in range(3):
                out |= p[dz:dz + a.shape[0], dy:dy + a.shape[1], dx:dx + a.shape[2]]
    return out
occd = dilate(occ)
def macro(a, m):
    Z, Y, X = [(s + m - 1) // m for s in a.shape]
    p = np.zeros((Z * m, Y * m, X * m), bool); p[:a.shape[0], :a.shape[1], :a.shape[2]] = a
    return p.reshape(Z, m, Y, m, X, m).any(axis=(1, 3, 5))
# camera (Serial/raytracer.cpp:130-161)
p = rt.default_params(W, H, **pkw)
pos = np.array(p.cam_pos[:], np.float64); tgt = np.array(p.cam_target[:], np.float64); up = np.array(p.cam_up[:], np.float64)
nrm = lambda v: v / np.linalg.norm(v)
wv = nrm(-(tgt - pos)); u = nrm(np.cross(nrm(up), wv)); v = nrm(np.cross(wv, u))
fd = 1.0 / (2.0 * np.tan(p.fov * np.pi / 360.0)); asp = W / H
ys, xs = np.meshgrid(np.arange(0, H, stride), np.arange(0, W, stride), indexing="ij")
xw = asp * (xs - W / 2.0 + 0.5) / W; yw = (ys - H / 2.0 + 0.5) / H
d = (-wv * fd)[None, None, :] + u[None, None, :] * xw[..., None] + v[None, None, :] * yw[..., None]
d = (d / np.linalg.norm(d, axis=-1, keepdims=True)).reshape(-1, 3)
with np.errstate(divide="ignore", invalid="ignore"):
    inv = 1.0 / d
    t0 = (lo - pos) * inv; t1 = (hi - pos) * inv
tn = np.minimum(t0, t1).max(axis=1); tf = np.maximum(t0, t1).min(axis=1)
enter = (tn <= tf) & (tf > 0)
tn = np.maximum(tn, 0.0)
R = d[enter]; T = tn[enter]
print(w, "sampled rays", len(d), "entering the bounds", len(R))
def walk(cells_occ, cell_w, ncell):
    """DDA through a grid of cell_w-wide cells; returns (steps, met_occupied)"""
    P = pos + R * T[:, None]
    c = np.clip(np.floor((P - lo) / cell_w).astype(np.int64), 0, ncell - 1)
    step = np.where(R >= 0, 1, -1)
    nxt = lo + (c + (R >= 0)) * cell_w
    with np.errstate(divide="ignore", invalid="ignore"):
        tnext = np.where(R != 0, (nxt - pos) / R, np.inf); dt = np.where(R != 0, cell_w / np.abs(R), np.inf)
    alive = np.ones(len(R), bool); steps = np.zeros(len(R), np.int64); met = np.zeros(len(R), bool)
    while alive.any():
        met |= alive & cells_occ[c[:, 2], c[:, 1], c[:, 0]]
        steps += alive
        ax = np.argmin(tnext, axis=1)
        idx = np.arange(len(R))
        c[idx, ax] += np.where(alive, step[idx, ax], 0)
        tnext[idx, ax] += dt[idx, ax]
        out = (c[idx, ax] < 0) | (c[idx, ax] >= ncell[ax])
        alive &= ~out
        c[idx, ax] = np.clip(c[idx, ax], 0, ncell[ax] - 1)
    return steps, met
steps, met = walk(occ, wd, n)
print("   fine walk to the grid exit: steps total %d, rays that meet an occupied voxel %d (%.1f %%), steps of the rays that meet none %d (%.1f %%)" % (
    steps.sum(), met.sum(), 100.0 * met.mean(), steps[~met].sum(), 100.0 * steps[~met].sum() / steps.sum()))
for m in (4, 8, 16):
    mo = macro(occd, m)
    ncell = np.array([mo.shape[2], mo.shape[1], mo.shape[0]])
    ms, mm = walk(mo, wd * m, ncell)
    assert not (met & ~mm).any(), "the macro test would drop a ray that meets an occupied voxel"
    print("   macro cells of %2d voxels: %4d of %5d flagged; macro steps per ray %.1f; rays kept %.1f %% (of entering), fine steps culled %.1f %% of all" % (
        m, mo.sum(), mo.size, ms.mean(), 100.0 * mm.mean(), 100.0 * steps[~mm].sum() / steps.sum()))
