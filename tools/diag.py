"""Scheduling diagnostics of the trace kernel for one workload (GPU)."""
import importlib, sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
rt = importlib.import_module("ray-tracer_amd")
for w in sys.argv[1:]:
    scene, pkw, W, H, label = bench.build_workload(rt, w)
    p = rt.default_params(W, H, **pkw)
    scene.render(p, want_rgb=False, want_ids=False, want_u8=True)
    scene.set_counting(True)
    scene.render(p, want_rgb=False, want_ids=False, want_u8=True)
    st = scene.last_stats()
    t = max(st["tiles_traced"], 1); r = max(st["sched_rounds"], 1)
    print(w, "ms_render(count)", round(st["ms_render"], 3), "tiles", st["tiles_traced"], "rounds/tile", round(st["sched_rounds"] / t, 1),
          "max rounds", st["tile_rounds_max"], "streams/round", round(st["sched_streams"] / r, 2), "cyc step/tile", int(st["sched_lanes"] / t), "cyc test/tile", int(st["cycles_test"] / t),
          "tile cycles avg", int(st["tile_cycles_sum"] / t), "max", st["tile_cycles_max"],
          "tests", st["tests_primary"] + st["tests_shadow"], "executed", st["tests_executed"], "steps", st["steps_primary"] + st["steps_shadow"],
          "| runs", st["runs_streamed"], "trips", st["run_ray_trips"], "cyc cull/tile", int(st["cycles_cull"] / t))
    print("   kernel span", st["span_cycles"], "busiest wave", st["wave_cycles_max"], "mean wave (4096)", int(st["tile_cycles_sum"] / 4096))
    nt = ((W + 7) // 8) * ((H + 7) // 8)
    c = scene.unit_costs(nt * 112)
    feat = c[16 * nt:20 * nt].astype(np.float64).reshape(4, nt); act = c[20 * nt:21 * nt].astype(np.float64)
    info = c[21 * nt:37 * nt]
    work_ = c[37 * nt:53 * nt]; ph1 = c[53 * nt:69 * nt]; ph2 = c[69 * nt:85 * nt]; t0s = c[85 * nt:101 * nt]
    c = c[:16 * nt]
    m = act > 0
    if m.sum() > 10:
        F, a = feat[:, m], act[m]
        for k, nm in enumerate(["primary tris", "primary voxels", "shadow tris", "shadow voxels"]):
            print("   corr(%s, measured) = %.3f" % (nm, np.corrcoef(F[k], a)[0, 1]))
        X = np.vstack([F, np.ones(len(a))]).T
        wts, *_ = np.linalg.lstsq(X, a, rcond=None)
        pred = X @ wts
        print("   least squares weights", np.round(wts, 1), "corr(fit, measured) = %.3f" % np.corrcoef(pred, a)[0, 1])
        np.save("gpurun_out/diag_%s_feat.npy" % w, feat); np.save("gpurun_out/diag_%s_act.npy" % w, act)
    n_units = int(np.flatnonzero(c)[-1]) + 1 if c.any() else 0
    q = c[:n_units].astype(np.float64)        # unit costs in queue order
    qi = info[:n_units]
    top = np.argsort(q)[::-1][:12]
    print("   heaviest units (cycles, slot, list, shift, tile cycles, tile estimate rank):")
    est_all = 1.0 * feat[0] + 500.0 * feat[1] + 700.0 * feat[3]
    order = np.argsort(np.argsort(-est_all))
    for u in top:
        t_ = int(qi[u] & 0xffffff)
        print("      %9d slot %6d list %2d shift %d tile %8d rank %d | tile (x,y)=(%d,%d) features %s | trips %d runs %d | step %dK cull %dK test %dK rounds %d" % (q[u], u, qi[u] >> 28, (qi[u] >> 24) & 15, act[t_], order[t_],
              t_ % ((W + 7) // 8), t_ // ((W + 7) // 8), feat[:, t_].astype(int).tolist(), work_[u] & 0x3ffff, work_[u] >> 18, (ph1[u] & 0xffff) >> 2, (ph1[u] >> 16) >> 2, (ph2[u] & 0xffffff) >> 2, ph2[u] >> 24))
    mm = act > 0
    print("      feature medians over listed tiles", np.median(feat[:, mm], axis=1).astype(int).tolist(), "p90", np.percentile(feat[:, mm], 90, axis=1).astype(int).tolist(), "max", feat[:, mm].max(axis=1).astype(int).tolist())
    for l in range(16):
        sel = (qi >> 28) == l
        if sel.any(): print("      list %2d: units %6d mean %8.0f max %8.0f" % (l, sel.sum(), q[sel].mean(), q[sel].max()))
    import heapq
    def makespan(costs, nw=4096):
        h = [0.0] * nw
        for x in costs:
            heapq.heapreplace(h, h[0] + x)
        return max(h)
    def latest(costs, lists_, nw=4096, top=8):
        h = [(0.0, i) for i in range(nw)]
        heapq.heapify(h)
        ends = []
        for j, x in enumerate(costs):
            t0_, wv = heapq.heappop(h)
            heapq.heappush(h, (t0_ + x, wv))
            ends.append((t0_ + x, t0_, x, int(lists_[j]), j))
        last_start = max(e[1] for e in ends)
        ends.sort(reverse=True)
        print("   simulated: last unit starts at %.0f; latest-ending units (end, start, cost, list, slot):" % last_start)
        for e in ends[:top]: print("      %.0f %.0f %.0f list %d slot %d" % e)
        fin = sorted(t for t, _ in h)
        print("   simulated wave end times: p10 %.0f p50 %.0f p90 %.0f p99 %.0f max %.0f" % (fin[len(fin)//10], fin[len(fin)//2], fin[len(fin)*9//10], fin[len(fin)*99//100], fin[-1]))
    latest(q, qi >> 28)
    # measured time line: units by start and end (clock wraps are ignored: a frame is far shorter than 2^32 cycles)
    raw = t0s[:n_units].astype(np.int64); ok_ = q > 0
    st_ = raw & 0xfffff; st_ = ((st_ - st_[0] + (1 << 19)) & 0xfffff) - (1 << 19); st_ = st_ - st_[ok_].min(); en_ = st_ + (raw >> 20) * 64
    T = en_[ok_].max()
    print("   measured (10 ns ticks): kernel %.0f; last unit starts at %.0f; units still running after the last start: %d" % (T, st_[ok_].max(), (en_[ok_] > st_[ok_].max()).sum()))
    for frac in (0.5, 0.6, 0.7, 0.8, 0.9):
        tcut = frac * T
        run = ok_ & (st_ <= tcut) & (en_ > tcut)
        ls = (qi >> 28)[run]
        print("      at %.0f%% of the kernel: %4d units running (lists 0-7: %d, 8-11: %d, 12-15: %d), started at t=0..5%%: %d" % (100 * frac, run.sum(), (ls < 8).sum(), ((ls >= 8) & (ls < 12)).sum(), (ls >= 12).sum(), (st_[run] < 0.05 * T).sum()))
    tr = (work_[:n_units] & 0x3ffff).astype(np.float64); it = (work_[:n_units] >> 18).astype(np.float64)
    X = np.vstack([tr, it, np.ones(n_units)]).T
    wts, *_ = np.linalg.lstsq(X, q, rcond=None)
    print("   unit cycles ~ %.0f * trips + %.0f * runs + %.0f (corr %.3f); totals: trips %.0f runs %.0f" % (wts[0], wts[1], wts[2], np.corrcoef(X @ wts, q)[0, 1], tr.sum(), it.sum()))
    print("   simulated greedy makespan: queue order %.0f | sorted descending %.0f | mean load %.0f | max unit %.0f" % (
        makespan(q), makespan(np.sort(q)[::-1]), q.sum() / 4096, q.max()))
    c = c[c > 0].astype(np.float64)
    if len(c):
        c.sort()
        tot = c.sum()
        print("   units", len(c), "cycles: mean %.0f p50 %.0f p90 %.0f p99 %.0f p99.9 %.0f max %.0f | share of total in top 1%% units: %.2f, top 0.1%%: %.2f" % (
            c.mean(), c[len(c)//2], c[int(len(c)*0.9)], c[int(len(c)*0.99)], c[int(len(c)*0.999)], c[-1], c[int(len(c)*0.99):].sum()/tot, c[int(len(c)*0.999):].sum()/tot))
        nw = 4096
        print("   ideal balanced = total/%d waves = %.0f cycles ; max unit = %.0f" % (nw, tot / nw, c[-1]))
