#!/usr/bin/env python3
"""Benchmark of the per-pixel render path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload NAME]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One step = one frame: on-GPU grid build (bounds, count, scan, scatter, order) + render (ray generation,
3D-DDA walk, triangle tests, shading, shadow rays) + framebuffer write (RGB8, the PPM body) and, for
N > 1, the RCCL gather of every rank's bands to rank 0 plus the band de-interleave there. The triangle
arrays are resident in HBM before the timed region. Metric: Mrays/s = (W*H primary rays + one shadow
ray per hit pixel) / frame time, whole job.

Workloads (BASELINE.json configs): "nefertiti4k" (default; configs[2]: the metric's configuration) renders
RT_NEFERTITI_OBJ when that file exists, else the labelled stand-in (blub subdivided 4x: 3.64 M triangles);
"spot1080" = configs[1]; "plane4k" = configs[4]; "main512" = the reference's main() scene.
With N > 1 the same frame is cut into interleaved 8-row bands (strong scaling).
"""
import argparse
import ctypes as C
import importlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
MESH = os.path.join(ROOT, "tests", "golden", "meshes")

# MI355X peaks (public spec figures; 256 CUs x 4 SIMD x 2.4 GHz)
PEAK_FP32 = 157.3e12
PEAK_FP64 = 78.6e12
PEAK_HBM = 8.0e12

# Algorithmic work per unit: triangle test = 9 FP32 + 57 FP64 flops + 3 FP64 divides as written in
# Triangle::Intersect; voxel step = 5 FP32 (1 add + 4 compares); ray = 60 FP32 flops + 9 FP32 divides
# (generation, slab test, DDA set-up) + 2 FP64 (sqrt, divide of normalize).
FLOP_TEST_F32, FLOP_TEST_F64 = 9, 60
FLOP_STEP_F32 = 5
FLOP_RAY_F32, FLOP_RAY_F64 = 69, 2


def build_workload(rt, name):
    """Returns (scene, params kwargs, W, H, label, v9 or mesh list for the oracle)."""
    s = rt.Scene()
    if name == "spot1080":
        s.load_mesh(os.path.join(MESH, "spot_triangulated.obj"))
        return s, {}, 1920, 1080, "spot_triangulated.obj 1920x1080"
    if name == "main512":
        s.load_mesh(os.path.join(MESH, "spot_triangulated.obj"))
        s.load_mesh(os.path.join(MESH, "blub_triangulated.obj"), True, (255, 0, 0), (1.5, 0, 0))
        return s, {}, 512, 512, "spot+blub (reference main()) 512x512"
    if name == "plane4k":
        path = "/tmp/rt_plane_spu100_%d.obj" % os.getpid()
        rt.make_plane_obj(path, 100)
        s.load_mesh(path)
        os.remove(path)
        return s, {}, 3840, 2160, "plane_mesh_creator SQUARES_PER_UNIT=100 (2,004,002 triangles) 3840x2160"
    if name in ("nefertiti4k", "nefertiti8k"):
        W, H = (3840, 2160) if name == "nefertiti4k" else (7680, 4320)
        real = os.environ.get("RT_NEFERTITI_OBJ", "")
        if real and os.path.exists(real):
            s.load_mesh(real, False)
            return s, dict(cam_pos=(0, -500, -100)), W, H, "nefertiti_triangulated.obj %dx%d" % (W, H)
        k = int(os.environ.get("RT_STANDIN_SUBDIV", "4"))
        tmp = rt.Scene()
        tmp.load_mesh(os.path.join(MESH, "blub_triangulated.obj"))
        v9 = rt.subdivide(tmp.triangles(), k)
        tmp.close()
        s.add_triangles(v9)
        return s, {}, W, H, "nefertiti STAND-IN (mesh absent): blub_triangulated.obj midpoint-subdivided %dx = %d triangles, %dx%d" % (k, v9.shape[0], W, H)
    raise SystemExit("unknown workload " + name)


def gather_parts(part, world, rank, out=None):
    """Gathers every rank's stacked bands to rank 0 (RCCL on GPUs; the same call runs over gloo in the CPU tests).
    Returns [world, rows, W, C] on rank 0 (written into `out` when given) and None elsewhere."""
    import torch
    import torch.distributed as dist
    if rank == 0 and out is None:
        out = torch.empty((world,) + tuple(part.shape), dtype=part.dtype, device=part.device)
    dist.gather(part, list(out.unbind(0)) if rank == 0 else None, dst=0)
    return out


def cpu_baseline(rt, scene, pkw, W, H, budget_s=12.0):
    """The oracle (C restatement of Serial/, bit-verified against the compiled reference) on ONE host core, in
    the reference's own full-walk mode, on a bounded sample of rows of the same frame."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    o = O.OracleScene()
    o.add_triangles(scene.triangles())
    grid_ms = o.build_grid()
    p = O.default_params(W, H, walk_mode=0, **pkw)
    rows_done, t_ms, rays = 0, 0.0, 0
    stride = max(1, H // 16)
    offsets = list(range(0, stride))
    cnt = O.Counters()
    t_start = time.time()
    # rows y = off, off+stride, ... : passes add more evenly spread rows until the time budget is used
    for off in offsets:
        for y in range(off, H, stride):
            before = cnt.hits
            _, _, ms = o.render(p, y0=y, y1=y + 1, want_rgb=True, want_ids=False, counters=cnt)
            t_ms += ms
            rows_done += 1
            rays += W + (cnt.hits - before)
        if time.time() - t_start > budget_s:
            break
    frac = rows_done / H
    est_frame_ms = grid_ms + t_ms / frac
    est_rays = rays / frac
    return {
        "value": round(est_rays / est_frame_ms / 1e3, 4), "unit": "Mrays/s", "cores": 1, "kind": "port",
        "sample": "%d of %d rows (every %dth row, %d passes) of the same frame, full-walk like the reference; "
                  "frame time extrapolated = grid build %.1f ms + rows %.1f ms / %.4f" % (rows_done, H, stride, rows_done * stride // H + 1, grid_ms, t_ms, frac),
        "est_frame_ms": round(est_frame_ms, 1),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default=os.environ.get("RT_BENCH_WORKLOAD", "nefertiti4k"))
    ap.add_argument("--walk", default="early", choices=["early", "full"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-rebuild", action="store_true", help="reuse the grid (render-only frames)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
        args.gpus = world
    # RT_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: the ranks share the visible GPUs
    # and the gather is staged through host memory. Its numbers mean nothing; it exercises the N > 1 code path.
    rehearsal = os.environ.get("RT_BENCH_BACKEND", "nccl") == "gloo"
    if rehearsal:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))

    rt = importlib.import_module("ray-tracer_amd")
    scene, pkw, W, H, label = build_workload(rt, args.workload)
    walk = rt.RT_WALK_FULL if args.walk == "full" else rt.RT_WALK_EARLY
    p = rt.default_params(W, H, walk_mode=walk, part_index=rank, part_count=world, **pkw)
    rows = rt.part_rows(p)
    dev = torch.device("cuda", local_rank)
    part_u8 = torch.empty((rows, W, 3), dtype=torch.uint8, device=dev)
    frame_u8 = torch.empty((H, W, 3), dtype=torch.uint8, device=dev) if rank == 0 else None
    gathered = torch.empty((world, rows, W, 3), dtype=torch.uint8, device=dev) if (world > 1 and rank == 0) else None
    stream = torch.cuda.current_stream().cuda_stream
    scene.upload()          # triangles resident in HBM before anything is timed

    def step():
        scene.render_device(p, rebuild_grid=not args.no_rebuild, d_u8=part_u8.data_ptr(), stream=stream)
        if world > 1:
            if rehearsal:
                torch.cuda.synchronize()
                g = gather_parts(part_u8.cpu(), world, rank)
                if rank == 0:
                    gathered.copy_(g)
            else:
                gather_parts(part_u8, world, rank, out=gathered)
            if rank == 0:
                rt.deinterleave_device(p, gathered.data_ptr(), frame_u8.data_ptr(), 3, stream=stream)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3

    # per-kernel device times of the timed frames (hipEvents recorded by the library on this stream)
    n_ev = min(args.steps, 120)
    st = [scene.stats_at(i) for i in range(n_ev)]
    ms_render = float(np.mean([x["ms_render"] for x in st]))
    ms_build = float(np.mean([x["ms_build"] for x in st]))
    hits_local = st[0]["hits"]
    rays_local = st[0]["primary_rays"] + hits_local

    # N > 1: the gathered, de-interleaved frame must be the frame one GPU renders alone (untimed check on rank 0)
    frame_check = None
    if world > 1 and rank == 0:
        p1 = rt.default_params(W, H, walk_mode=walk, part_index=0, part_count=1, **pkw)
        single = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
        scene.render_device(p1, rebuild_grid=False, d_u8=single.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        frame_check = bool(torch.equal(single, frame_u8))
        del single

    # one extra, untimed frame with the counting kernel: voxel steps and triangle tests actually needed
    scene.set_counting(True)
    scene.render_device(p, rebuild_grid=False, d_u8=part_u8.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    cs = scene.last_stats()
    scene.set_counting(False)

    tot = torch.tensor([rays_local, hits_local], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    rays_total, hits_total = float(tot[0].item()), float(tot[1].item())

    if rank == 0:
        tests_ref = cs["tests_primary"] + cs["tests_shadow"]      # tests the reference's order needs (early exit)
        tests = cs["tests_executed"]                              # tests run: runs whose box the line misses are skipped
        steps_ = cs["steps_primary"] + cs["steps_shadow"]
        nrays = cs["primary_rays"] + cs["hits"]
        f32 = tests * FLOP_TEST_F32 + steps_ * FLOP_STEP_F32 + nrays * FLOP_RAY_F32
        f64 = tests * FLOP_TEST_F64 + nrays * FLOP_RAY_F64
        t_render = ms_render * 1e-3
        t_peak = f32 / PEAK_FP32 + f64 / PEAK_FP64
        # compulsory HBM traffic of the render kernel: records 48 B/ref, cell table, RGB8 framebuffer
        hbm_bytes = cs["n_refs"] * 48 + (cs["n_cells"] + 1) * 4 + cs["n_cells"] / 8 + cs["primary_rays"] * 3
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tinfo = json.load(open(tpath)).get(args.workload)
                traffic = tinfo["bytes_per_launch"] if tinfo else None
            except Exception:
                traffic = None
        out = {
            "metric": {"nefertiti4k": "Mrays/s (primary+shadow), nefertiti 4K", "nefertiti8k": "Mrays/s (primary+shadow), nefertiti 8K"}.get(args.workload, "Mrays/s (primary+shadow)"),
            "value": round(rays_total / (ms_per_step * 1e-3) / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "strong",
            "frame_check": frame_check,
            "vs_baseline": None,
            "dtype": "f32+f64",
            "data": "synthetic" if "STAND-IN" in label or "plane" in label else "reference meshes",
            "config": {"workload": label, "walk": args.walk, "rays_per_frame": int(rays_total), "hits": int(hits_total),
                       "frame": "grid build + render + RGB8 framebuffer" + (" + RCCL gather + de-interleave" if world > 1 else ""),
                       "partition": "interleaved 8-row bands, scene replicated" if world > 1 else "single GPU",
                       "triangles": cs["n_triangles"], "grid": cs["grid_n"], "cell_refs": cs["n_refs"]},
            "frame_ms": {"build": round(ms_build, 4), "render": round(ms_render, 4), "device_total": round(ms_build + ms_render, 4)},
            "render_only_mrays": round(rays_local / t_render / 1e6, 2) if t_render > 0 else None,
            "roofline": {
                "bound": "valu", "kernel": "k_trace",
                "achieved": round((f32 + f64) / t_render / 1e12, 4), "peak": round((f32 + f64) / t_peak / 1e12, 2), "unit": "TFLOP/s",
                "frac": round(t_peak / t_render, 5),
                "traffic": traffic,
                "traffic_note": "HBM-side bytes of one k_trace launch from the L2 fabric counters (profiles/traffic.json: separate rocprofv3 --pmc passes, FETCH_SIZE doubled on gfx950)",
                "note": "flops of the triangle tests k_trace executes (%d; the reference's early-exit order needs %d, the rest are skipped by the per-run "
                        "box test), %d voxel steps and %d rays of rank 0's frame, over the mean k_trace time of the timed frames (hipEvents around the launch, recorded by the library on its stream); "
                        "peak = the same flops at 157.3 TF fp32 + 78.6 TF fp64" % (tests, tests_ref, steps_, nrays),
            },
            "roofline_hbm": {
                "bound": "hbm", "achieved": round(hbm_bytes / t_render / 1e9, 2), "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                "frac": round(hbm_bytes / t_render / PEAK_HBM, 6), "traffic": traffic,
                "note": "compulsory bytes (48 B per cell reference + cell table + RGB8 frame); the path is VALU-issue bound, not HBM bound",
            },
        }
        if not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(rt, scene, pkw, W, H)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
