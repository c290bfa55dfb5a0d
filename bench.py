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
# Traversal stress: the same scene seen from close by, so that most primary rays enter the grid and walk a deep, crowded part
# of it (Serial camera: 13 % of the stand-in's rays enter, 4 % hit; this one: 77 % enter, 31 % hit). Not the headline frame.
STRESS_CAMERA = dict(cam_pos=(1.0, 1.6, 1.0), cam_target=(0.0, 0.2, -0.45))

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
    if name in ("nefertiti4k", "nefertiti8k", "nefertiti4k_close"):
        W, H = (7680, 4320) if name == "nefertiti8k" else (3840, 2160)
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
        label = "nefertiti STAND-IN (mesh absent): blub_triangulated.obj midpoint-subdivided %dx = %d triangles, %dx%d" % (k, v9.shape[0], W, H)
        if name == "nefertiti4k_close":      # the traversal-stress camera as a workload of its own (profiling)
            return s, dict(STRESS_CAMERA), W, H, label + ", close camera %s" % (STRESS_CAMERA["cam_pos"],)
        return s, {}, W, H, label
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


def cpu_baseline(rt, scene, pkw, W, H, walk, budget_s=12.0):
    """The oracle (C restatement of Serial/, bit-verified against the compiled reference) on ONE host core, in
    the reference's own full-walk mode, on a bounded sample of rows of the same frame. The same rows of the frame the
    GPU path renders in the benchmarked walk mode are compared with them pixel by pixel (second return value)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    o = O.OracleScene()
    o.add_triangles(scene.triangles())
    grid_ms = o.build_grid()
    p = O.default_params(W, H, walk_mode=0, **pkw)
    g_rgb, _, g_ids = scene.render(rt.default_params(W, H, walk_mode=walk, **pkw))
    other = rt.RT_WALK_EARLY if walk == rt.RT_WALK_FULL else rt.RT_WALK_FULL
    h_rgb, _, h_ids = scene.render(rt.default_params(W, H, walk_mode=other, **pkw))       # the other walk mode, same rows
    n_px = n_id = n_col = n_bits = 0
    m_id = m_col = m_bits = 0
    rows_done, t_ms, rays = 0, 0.0, 0
    stride = max(1, H // 16)
    offsets = list(range(0, stride))
    cnt = O.Counters()
    t_start = time.time()
    # rows y = off, off+stride, ... : passes add more evenly spread rows until the time budget is used
    for off in offsets:
        for y in range(off, H, stride):
            before = cnt.hits
            o_rgb, o_ids, ms = o.render(p, y0=y, y1=y + 1, want_rgb=True, want_ids=True, counters=cnt)
            t_ms += ms
            d = np.abs(g_rgb[y:y + 1].astype(np.float64) - o_rgb.astype(np.float64)).max(axis=-1)
            n_px += W
            n_id += int((g_ids[y:y + 1] != o_ids).sum())
            n_col += int((np.where(np.isnan(d), np.inf, d) > 1.0).sum())
            n_bits += int((g_rgb[y:y + 1].view(np.uint32) != o_rgb.view(np.uint32)).any(axis=-1).sum())
            d2 = np.abs(h_rgb[y:y + 1].astype(np.float64) - o_rgb.astype(np.float64)).max(axis=-1)
            m_id += int((h_ids[y:y + 1] != o_ids).sum())
            m_col += int((np.where(np.isnan(d2), np.inf, d2) > 1.0).sum())
            m_bits += int((h_rgb[y:y + 1].view(np.uint32) != o_rgb.view(np.uint32)).any(axis=-1).sum())
            rows_done += 1
            rays += W + (cnt.hits - before)
        if time.time() - t_start > budget_s:
            break
    frac = rows_done / H
    est_frame_ms = grid_ms + t_ms / frac
    est_rays = rays / frac
    parity = {"against": "oracle, full walk (what the reference executes)", "gpu_walk": "full" if walk == rt.RT_WALK_FULL else "early",
              "pixels_checked": n_px, "rows_checked": rows_done, "id_mismatch": n_id, "colour_gt_1_255": n_col, "colour_bits_differ": n_bits,
              "agreement": round(1.0 - n_col / max(n_px, 1), 7),
              "other_walk": {"gpu_walk": "full" if other == rt.RT_WALK_FULL else "early", "pixels_checked": n_px, "id_mismatch": m_id,
                             "colour_gt_1_255": m_col, "colour_bits_differ": m_bits}}
    # The reference ITSELF (Serial/raytracer.cpp compiled from its own sources into oracle/_ref/libref.so, where that file
    # was built) beside the port on the reference's own CPU-runnable case, configs[0]: spot at 640x480, same core. This
    # says what the port's number is worth as a stand-in for the reference on the large frame above.
    ref_check = None
    if O.ref_available():
        spot = os.path.join(MESH, "spot_triangulated.obj")
        r = O.RefScene()
        r.add_obj(spot)
        r_grid = r.build_grid()
        r_rgb, _, r_ms = r.render(640, 480, want_ids=False)
        o2 = O.OracleScene()
        o2.add_obj(spot)
        o_grid = o2.build_grid()
        c2 = O.Counters()
        o_rgb2, _, o_ms = o2.render(O.default_params(640, 480, walk_mode=0), want_ids=False, counters=c2)
        n_rays = 640 * 480 + c2.hits
        ref_check = {"workload": "configs[0]: spot_triangulated.obj 640x480, Serial camera, %d rays" % n_rays, "cores": 1,
                     "reference": {"kind": "reference", "value": round(n_rays / (r_grid + r_ms) / 1e3, 4), "unit": "Mrays/s",
                                   "frame_ms": round(r_grid + r_ms, 1), "grid_ms": round(r_grid, 1)},
                     "port": {"kind": "port", "value": round(n_rays / (o_grid + o_ms) / 1e3, 4), "unit": "Mrays/s",
                              "frame_ms": round(o_grid + o_ms, 1), "grid_ms": round(o_grid, 1)},
                     "images_bit_equal": bool(np.array_equal(r_rgb.view(np.uint32), o_rgb2.view(np.uint32)))}
    port = {"value": round(est_rays / est_frame_ms / 1e3, 4), "unit": "Mrays/s", "cores": 1, "kind": "port", "est_frame_ms": round(est_frame_ms, 1)}
    # The reference ITSELF on the benchmarked frame: the scene's triangles go through an OBJ file its own loader reads back
    # bit for bit, its own GridAccel is built, and evenly spread rows of the frame are rendered by its fast_trace (about 4 s
    # worth); the rows are compared with the oracle's. This is the headline CPU number where libref.so exists.
    if O.ref_available() and all(k in ("cam_pos", "cam_target") for k in pkw):
        try:
            tri = scene.triangles()
            path = "/tmp/rt_bench_scene_%d.obj" % os.getpid()
            t_w = time.time()
            O.write_obj(path, tri)
            t_w = time.time() - t_w
            r = O.RefScene()
            t_l = time.time()
            n_loaded = r.add_obj(path, False)
            t_l = time.time() - t_l
            os.remove(path)
            tris_equal = n_loaded == tri.shape[0] and bool(np.array_equal(r.triangles().view(np.uint32), tri.view(np.uint32)))
            r_grid = r.build_grid()
            cam = tuple(pkw.get("cam_pos", (3, 5, 3))) + tuple(pkw.get("cam_target", (0, 0, 0))) + (0, -1, 0, 45)
            per_row = t_ms / max(rows_done, 1)
            stride_r = max(1, int(np.ceil(1.4 * per_row * H / 4000.0)))
            r_ms, r_rows, r_rays, rows_equal = 0.0, 0, 0, True
            for y in range(stride_r // 2, H, stride_r):
                c1 = O.Counters()
                o_rgb, _, _ = o.render(p, y0=y, y1=y + 1, want_rgb=True, want_ids=False, counters=c1)
                r_rgb, _, ms = r.render(W, H, y0=y, y1=y + 1, cam=cam, want_ids=False)
                r_ms += ms; r_rows += 1; r_rays += W + c1.hits
                rows_equal = rows_equal and bool(np.array_equal(r_rgb.view(np.uint32), o_rgb.view(np.uint32)))
            r_frac = r_rows / H
            r_frame_ms = r_grid + r_ms / r_frac
            return parity, {
                "value": round((r_rays / r_frac) / r_frame_ms / 1e3, 4), "unit": "Mrays/s", "cores": 1, "kind": "reference",
                "sample": "Serial/raytracer.cpp itself (oracle/_ref/libref.so, compiled from the reference's sources): its load_mesh on the scene written as OBJ "
                          "(%.1f s to write, %.1f s to load, triangles bit-equal: %s), its GridAccel (%.1f ms), its fast_trace on %d of %d rows (every %dth row) of the "
                          "same frame; frame time extrapolated = grid build + rows %.1f ms / %.4f" % (t_w, t_l, tris_equal, r_grid, r_rows, H, stride_r, r_ms, r_frac),
                "est_frame_ms": round(r_frame_ms, 1),
                "rows_bit_equal_to_oracle": rows_equal,
                "port": dict(port, sample="the oracle (C restatement) on %d of %d rows of the same frame, full-walk like the reference; frame time extrapolated = "
                                          "grid build %.1f ms + rows %.1f ms / %.4f" % (rows_done, H, grid_ms, t_ms, frac)),
                "reference_check": ref_check,
            }
        except Exception as e:          # the reference leg is an extra: without it the port's number stands
            port["reference_error"] = repr(e)
    return parity, {
        "value": port["value"], "unit": "Mrays/s", "cores": 1, "kind": "port",
        "reference_check": ref_check, "reference_error": port.get("reference_error"),
        "sample": "%d of %d rows (every %dth row, %d passes) of the same frame, full-walk like the reference; "
                  "frame time extrapolated = grid build %.1f ms + rows %.1f ms / %.4f" % (rows_done, H, stride, rows_done * stride // H + 1, grid_ms, t_ms, frac),
        "est_frame_ms": round(est_frame_ms, 1),
    }


def launch_command(n, argv, port=None):
    """The command `bench.py --gpus n` runs itself under when no launcher has set WORLD_SIZE."""
    port = port or os.environ.get("MASTER_PORT") or str(29500 + os.getpid() % 2000)
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def launch_ranks(n, argv):
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    r = subprocess.run(launch_command(n, argv), env=env)
    return r.returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--workload", default=os.environ.get("RT_BENCH_WORKLOAD", "nefertiti4k"))
    ap.add_argument("--walk", default="early", choices=["early", "full"])
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-rebuild", action="store_true", help="reuse the grid (render-only frames)")
    ap.add_argument("--in-flight", type=int, default=int(os.environ.get("RT_BENCH_IN_FLIGHT", "4")),
                    help="frames in flight: frame k is issued on stream k %% D, each stream with its own working set in the scene (1 = one frame at a time)")
    ap.add_argument("--no-extras", action="store_true", help="headline frames only: no host-writeback loop, no stress camera (profiling runs: every "
                                                                "launch of the trace kernel is then one of the timed frame's)")
    args = ap.parse_args()

    # `python bench.py --gpus N` by itself: start the N ranks as a CHILD process (one rank per GPU under torch.distributed.run),
    # relay its output and return its exit code. This happens before torch is imported or the GPU is touched; the program
    # is never replaced (no exec).
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        args.gpus = world
    # RT_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks: the ranks share the visible GPUs
    # and the gather is staged through host memory. Its numbers mean nothing; it exercises the N > 1 code path.
    rehearsal = os.environ.get("RT_BENCH_BACKEND", "nccl") == "gloo"
    if rehearsal:
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    # RT_BENCH_FORCE_EXCHANGE=1: run the exchange step (process group, gather, de-interleave, frame check) with one rank too,
    # to exercise that code path on a box with a single GPU. Its numbers are those of a one-GPU frame plus a no-op gather.
    multi = world > 1 or os.environ.get("RT_BENCH_FORCE_EXCHANGE", "") == "1"
    if multi:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
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
    stream = torch.cuda.current_stream().cuda_stream
    scene.upload()          # triangles resident in HBM before anything is timed
    scene.set_estimate_reuse(False)   # every timed frame does all of its work, the tile cost estimate included (see estimate_reused below)

    # Frames in flight: a frame is a dozen dependent kernels, the first ones short and narrow, the last one ending in a tail
    # of a few long wavefronts, so one frame at a time leaves much of the GPU idle. Frame k is issued on lane k % D: a
    # stream of its own (checked by the library to run beside the others) with its own working set in the scene and its own
    # output buffers; everything of a frame, the exchange for N > 1 included (gather to rank 0 + band de-interleave there),
    # is ordered by its lane's stream, and lanes overlap on the GPU. Every frame still builds its own grid.
    depth = max(1, min(args.in_flight, 8))
    scene.set_frames_in_flight(depth)
    lane_handles = rt.create_streams(depth)
    lanes = [torch.cuda.ExternalStream(h, device=dev) for h in lane_handles]
    n_lanes = len(lanes)
    part_bufs = [torch.empty((rows, W, 3), dtype=torch.uint8, device=dev) for _ in lanes]
    part_u8 = part_bufs[0]
    frame_bufs = [torch.empty((H, W, 3), dtype=torch.uint8, device=dev) for _ in lanes] if rank == 0 else None
    gathered = [torch.empty((world, rows, W, 3), dtype=torch.uint8, device=dev) for _ in lanes] if (multi and rank == 0) else None
    torch.cuda.synchronize()

    def step(k):
        b = k % n_lanes
        scene.render_device(p, rebuild_grid=not args.no_rebuild, d_u8=part_bufs[b].data_ptr(), stream=lane_handles[b])
        if multi:
            with torch.cuda.stream(lanes[b]):
                if rehearsal:
                    lanes[b].synchronize()
                    g = gather_parts(part_bufs[b].cpu(), world, rank)
                    if rank == 0:
                        gathered[b].copy_(g)
                else:
                    gather_parts(part_bufs[b], world, rank, out=gathered[b] if rank == 0 else None)
                if rank == 0:
                    rt.deinterleave_device(p, gathered[b].data_ptr(), frame_bufs[b].data_ptr(), 3, stream=lane_handles[b])

    def sync_all():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
        torch.cuda.synchronize()

    # Two side measurements first (untimed by the driver; they also bring the GPU to its working clocks before the timed
    # region: with 5 warm-up frames only, the same 20 timed frames are 5 % slower).
    # (1) the frame alone on the GPU in the benchmarked configuration: a few frames one after the other on one lane
    ms_render_alone = ms_build_alone = None
    if n_lanes > 1:
        n_alone = max(3, min(args.steps, 10))
        for _ in range(n_alone):
            scene.render_device(p, rebuild_grid=not args.no_rebuild, d_u8=part_bufs[0].data_ptr(), stream=lane_handles[0])
        torch.cuda.synchronize()
        st_a = [scene.stats_at(i) for i in range(n_alone - 1)]
        ms_render_alone = float(np.mean([x["ms_render"] for x in st_a]))
        ms_build_alone = float(np.mean([x["ms_build"] for x in st_a]))
    # (2) frame time when latency is what counts: the scene set up for one frame at a time (one working set; the trace kernel
    # then takes the whole GPU and large meshes prepare their work list beside the grid build), frames issued back to back
    # on one stream
    one_at_a_time = None
    if world == 1:
        scene.set_frames_in_flight(1)
        n_one = max(4, min(args.steps, 20))
        for _ in range(2):
            scene.render_device(p, rebuild_grid=not args.no_rebuild, d_u8=part_bufs[0].data_ptr(), stream=lane_handles[0])
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(n_one):
            scene.render_device(p, rebuild_grid=not args.no_rebuild, d_u8=part_bufs[0].data_ptr(), stream=lane_handles[0])
        torch.cuda.synchronize()
        dt_one = (time.perf_counter() - t1) / n_one
        st_o = [scene.stats_at(i) for i in range(n_one)]
        one_at_a_time = {"ms_per_frame": round(dt_one * 1e3, 4), "frames": n_one,
                         "build": round(float(np.mean([x["ms_build"] for x in st_o])), 4), "render": round(float(np.mean([x["ms_render"] for x in st_o])), 4)}
        scene.set_frames_in_flight(depth)
        for k in range(n_lanes):          # the working sets of the lanes exist again before anything is timed
            scene.render_device(p, rebuild_grid=not args.no_rebuild, d_u8=part_bufs[k].data_ptr(), stream=lane_handles[k])
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    sync_all()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    sync_all()
    dt = time.perf_counter() - t0
    if multi:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3

    # device times of the timed frames (clock values the frame's kernels write). With frames in flight a frame shares the
    # GPU with its neighbours: its build and trace kernel take longer than alone, and their sum is the frame's latency.
    n_ev = min(args.steps, 120)
    st = [scene.stats_at(i) for i in range(n_ev)]
    ms_render = float(np.mean([x["ms_render"] for x in st]))
    ms_build = float(np.mean([x["ms_build"] for x in st]))
    hits_local = st[0]["hits"]
    rays_local = st[0]["primary_rays"] + hits_local
    if ms_render_alone is None:
        ms_render_alone, ms_build_alone = ms_render, ms_build
    if one_at_a_time is not None:
        one_at_a_time["mrays"] = round(rays_local / (one_at_a_time["ms_per_frame"] * 1e-3) / 1e6, 2)
    # N > 1: the gathered, de-interleaved frame must be the frame one GPU renders alone (untimed check on rank 0)
    frame_check = None
    if multi and rank == 0:
        p1 = rt.default_params(W, H, walk_mode=walk, part_index=0, part_count=1, **pkw)
        single = torch.empty((H, W, 3), dtype=torch.uint8, device=dev)
        scene.render_device(p1, rebuild_grid=False, d_u8=single.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        frame_check = bool(torch.equal(single, frame_bufs[(args.steps - 1) % n_lanes]))
        del single

    # one extra, untimed frame with the counting kernel: voxel steps and triangle tests actually needed
    scene.set_counting(True)
    scene.render_device(p, rebuild_grid=False, d_u8=part_u8.data_ptr(), stream=stream)
    torch.cuda.synchronize()
    cs = scene.last_stats()
    scene.set_counting(False)

    # Frame including the writeback to a host-visible buffer (Serial/raytracer.cpp:141,178-185 ends with the image in host
    # memory): RGB8 into pinned host memory, copied on a stream of its own, so that frame k's copy runs beside frame k+1's
    # grid build; two device and two host buffers. Single GPU only (on N GPUs rank 0 holds the gathered frame).
    wb = None
    if world == 1 and not args.no_extras:
        n_buf = n_lanes
        # the copy of a frame follows its trace on the frame's lane, beside the other lanes' frames
        # small frames: a copy kernel that writes the mapped host buffer; large ones: hipMemcpyAsync (measured on the test
        # box: the DMA path has about 0.5 ms of fixed cost per copy but the higher rate, 21 against 17 GB/s)
        use_kernel = os.environ.get("RT_WB_COPY", "kernel" if rows * W * 3 < (12 << 20) else "memcpy") == "kernel"
        d_buf = part_bufs
        h_buf = [torch.empty((rows, W, 3), dtype=torch.uint8, pin_memory=True) for _ in range(n_buf)]
        p_copy = rt.default_params(W, rows)                          # whole-buffer copy: one part, rows as they are

        def step_wb(k):
            b = k % n_buf
            scene.render_device(p, rebuild_grid=not args.no_rebuild, d_u8=d_buf[b].data_ptr(), stream=lane_handles[b])
            if use_kernel:
                # pinned host memory is mapped into the device's address space: a copy kernel writes it over PCIe
                rt.deinterleave_device(p_copy, d_buf[b].data_ptr(), h_buf[b].data_ptr(), 3, stream=lane_handles[b])
            else:
                with torch.cuda.stream(lanes[b]):
                    h_buf[b].copy_(d_buf[b], non_blocking=True)

        for k in range(max(args.warmup, n_buf)):
            step_wb(k)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            step_wb(k)
        torch.cuda.synchronize()
        dt_wb = (time.perf_counter() - t0) / args.steps
        last = (args.steps - 1) % n_buf
        wb = {"ms": round(dt_wb * 1e3, 4), "bytes_equal": bool(torch.equal(h_buf[last], d_buf[last].cpu())),
              "mrays": round(rays_local / dt_wb / 1e6, 2), "bytes_per_frame": int(h_buf[0].numel()),
              "how": "%d lanes, each with a device and a pinned host buffer; %s behind the trace on the frame's lane, beside the other lanes' frames" % (n_buf, "copy kernel writing the mapped host buffer" if use_kernel else "hipMemcpyAsync")}

    # Traversal stress beside the headline: same scene, close camera (untimed by the driver; a few frames)
    stress = None
    if world == 1 and not args.no_extras and args.workload in ("nefertiti4k", "nefertiti8k") and "cam_pos" not in pkw:
        ps = rt.default_params(W, H, walk_mode=walk, **STRESS_CAMERA)
        n_s = max(3, min(args.steps, 10))
        for k in range(n_lanes):
            scene.render_device(ps, rebuild_grid=not args.no_rebuild, d_u8=part_bufs[k].data_ptr(), stream=lane_handles[k])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n_s):
            scene.render_device(ps, rebuild_grid=not args.no_rebuild, d_u8=part_bufs[k % n_lanes].data_ptr(), stream=lane_handles[k % n_lanes])
        torch.cuda.synchronize()
        dt_s = (time.perf_counter() - t0) / n_s
        # kernel times of this frame alone on the GPU: a few frames one after the other on one lane
        for _ in range(4):
            scene.render_device(ps, rebuild_grid=not args.no_rebuild, d_u8=part_bufs[0].data_ptr(), stream=lane_handles[0])
        torch.cuda.synchronize()
        st_s = [scene.stats_at(i) for i in range(3)]
        ms_r = float(np.mean([x["ms_render"] for x in st_s])); ms_b = float(np.mean([x["ms_build"] for x in st_s]))
        scene.set_counting(True)
        scene.render_device(ps, rebuild_grid=False, d_u8=part_u8.data_ptr(), stream=stream)
        torch.cuda.synchronize()
        c2 = scene.last_stats()
        scene.set_counting(False)
        nr = c2["primary_rays"] + c2["hits"]
        f32s = c2["tests_executed"] * FLOP_TEST_F32 + (c2["steps_primary"] + c2["steps_shadow"]) * FLOP_STEP_F32 + nr * FLOP_RAY_F32
        f64s = c2["tests_executed"] * FLOP_TEST_F64 + nr * FLOP_RAY_F64
        stress = {"workload": "same scene, camera %s -> %s (%.0f %% of the primary rays enter the grid, %.0f %% hit)" % (
                      STRESS_CAMERA["cam_pos"], STRESS_CAMERA["cam_target"], 100.0 * c2["primary_entering"] / c2["primary_rays"], 100.0 * c2["hits"] / c2["primary_rays"]),
                  "value": round(nr / dt_s / 1e6, 2), "unit": "Mrays/s", "ms_per_frame": round(dt_s * 1e3, 4), "frames": n_s, "frames_in_flight": n_lanes,
                  "frame_ms_alone": {"build": round(ms_b, 4), "render": round(ms_r, 4)},
                  "ns_per_entering_ray": round(ms_r * 1e6 / max(c2["primary_entering"] + c2["shadow_entering"], 1), 3),
                  "rays": int(nr), "entering": int(c2["primary_entering"] + c2["shadow_entering"]), "tests_executed": int(c2["tests_executed"]),
                  "roofline_frac": round((f32s / PEAK_FP32 + f64s / PEAK_FP64) / (ms_r * 1e-3), 5)}

    # The reference-faithful mode beside the headline: every primary ray walked to the grid exit like Serial/grid.h:208-228
    # (the headline frames stop a primary ray once its nearest hit lies before the next voxel crossing). Same pipeline, fewer frames.
    full_walk = None
    if world == 1 and not args.no_extras and args.walk == "early":
        pf = rt.default_params(W, H, walk_mode=rt.RT_WALK_FULL, **pkw)
        n_f = max(4, min(args.steps, 40))
        for k in range(n_lanes):
            scene.render_device(pf, rebuild_grid=not args.no_rebuild, d_u8=part_bufs[k].data_ptr(), stream=lane_handles[k])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n_f):
            scene.render_device(pf, rebuild_grid=not args.no_rebuild, d_u8=part_bufs[k % n_lanes].data_ptr(), stream=lane_handles[k % n_lanes])
        torch.cuda.synchronize()
        dt_f = (time.perf_counter() - t0) / n_f
        scene.set_frames_in_flight(1)
        for _ in range(6):
            scene.render_device(pf, rebuild_grid=not args.no_rebuild, d_u8=part_bufs[0].data_ptr(), stream=lane_handles[0])
        torch.cuda.synchronize()
        st_f = [scene.stats_at(i) for i in range(4)]
        scene.set_frames_in_flight(depth)
        full_walk = {"what": "every primary ray walked to the grid exit, as the reference does (Serial/grid.h:208-228)",
                     "value": round(rays_local / dt_f / 1e6, 2), "unit": "Mrays/s", "ms_per_frame": round(dt_f * 1e3, 4), "frames": n_f, "frames_in_flight": n_lanes,
                     "one_at_a_time_ms": {"build": round(float(np.mean([x["ms_build"] for x in st_f])), 4), "render": round(float(np.mean([x["ms_render"] for x in st_f])), 4)}}

    # Every frame above ran its own tile cost estimate. The frames of this benchmark all have one camera and one triangle list,
    # which is the case the library's reuse of the estimate is for (rt_scene_set_estimate_reuse: from its second frame on a lane
    # takes the estimate of its frame before; scheduling only, no pixel depends on it; the grid is still built per frame). The
    # same loop with it, as a side figure:
    reused = None
    if world == 1 and not args.no_extras:
        scene.set_estimate_reuse(True)
        n_e = max(4, min(args.steps, 40))
        for k in range(2 * n_lanes):
            scene.render_device(p, rebuild_grid=not args.no_rebuild, d_u8=part_bufs[k % n_lanes].data_ptr(), stream=lane_handles[k % n_lanes])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(n_e):
            scene.render_device(p, rebuild_grid=not args.no_rebuild, d_u8=part_bufs[k % n_lanes].data_ptr(), stream=lane_handles[k % n_lanes])
        torch.cuda.synchronize()
        dt_e = (time.perf_counter() - t0) / n_e
        scene.set_frames_in_flight(1)
        for _ in range(3):
            scene.render_device(p, rebuild_grid=not args.no_rebuild, d_u8=part_bufs[0].data_ptr(), stream=lane_handles[0])
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            scene.render_device(p, rebuild_grid=not args.no_rebuild, d_u8=part_bufs[0].data_ptr(), stream=lane_handles[0])
        torch.cuda.synchronize()
        dt_e1 = (time.perf_counter() - t0) / 10
        scene.set_frames_in_flight(depth)
        scene.set_estimate_reuse(False)
        reused = {"what": "the same pipeline with the tile cost estimate of a lane's previous frame reused (camera, light and triangles unchanged between frames)",
                  "value": round(rays_local / dt_e / 1e6, 2), "unit": "Mrays/s", "ms_per_frame": round(dt_e * 1e3, 4), "frames": n_e,
                  "one_at_a_time_ms_per_frame": round(dt_e1 * 1e3, 4)}

    tot = torch.tensor([rays_local, hits_local], dtype=torch.float64, device=dev)
    if multi:
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
        t_render_alone = ms_render_alone * 1e-3
        # Time one launch costs the GPU. One frame at a time: the launch's duration. Frames in flight: launches overlap
        # and their durations add up to more than the wall time, so the wall time per frame is used, which charges the
        # launch with everything else in the frame too (grid build, work lists): a lower bound of the kernel's rate.
        # Duration of one launch of the dominant kernel, the figure a kernel trace shows: the launch as issued for one frame at a
        # time (the GPU to itself). With frames in flight the launches overlap and no single duration describes them; the wall
        # time per frame is then a throughput figure of the whole GPU and is reported as such, under its own name.
        t_wall = ms_per_step * 1e-3
        if one_at_a_time and one_at_a_time["render"] > 0:
            t_launch = one_at_a_time["render"] * 1e-3
            launch_note = ("mean start-to-end time of the k_trace launches of %d frames issued one at a time before the timed region (clock values the kernel "
                           "writes at its start and end; the average a rocprofv3 kernel trace of `bench.py --in-flight 1` shows, profiles/*_one_at_a_time_kernel_stats.csv)" % one_at_a_time["frames"])
        elif n_lanes > 1:
            t_launch = t_render_alone
            launch_note = "mean start-to-end time of the k_trace launches of frames issued one after the other on one stream (clock values the kernel writes)"
        else:
            t_launch = t_render
            launch_note = "mean start-to-end time of the k_trace launches of the timed region (clock values the kernel writes)"
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
            "metric": {"nefertiti4k": "Mrays/s (primary+shadow), nefertiti 4K", "nefertiti8k": "Mrays/s (primary+shadow), nefertiti 8K"}.get(args.workload, "Mrays/s (primary+shadow)")
                      + "; value = frames ending in HBM, value_host_visible = the same frames ending in a host-visible RGB8 buffer",
            "value": round(rays_total / (ms_per_step * 1e-3) / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            # the frame as Serial/raytracer.cpp:141,178-185 ends it: image in host memory (RGB8 copied to pinned host memory behind the
            # trace, same pipeline; bound by the PCIe rate of the box); None for N > 1 and with --no-extras
            "value_host_visible": wb["mrays"] if wb else None,
            "ms_per_step_host_visible": wb["ms"] if wb else None,
            # ms_per_step is wall time per completed frame with several frames in flight, not the time one frame takes:
            "frame_ms_latency": {"in_flight": round(ms_build + ms_render, 4), "one_at_a_time": one_at_a_time["ms_per_frame"] if one_at_a_time else None,
                                 "note": "first kernel to last kernel of one frame: as the timed frames saw it (sharing the GPU with the other frames in flight), and "
                                         "frames issued one at a time (wall time per frame)"},
            "full_walk": full_walk,
            "estimate_reused": reused,
            "higher_is_better": True,
            "scaling": "strong",
            "frame_check": frame_check,
            "vs_baseline": None,
            "dtype": "f32+f64",
            "data": "synthetic" if "STAND-IN" in label or "plane" in label else "reference meshes",
            "config": {"workload": label, "walk": args.walk, "rays_per_frame": int(rays_total), "hits": int(hits_total),
                       "frame": "grid build + render + RGB8 framebuffer" + (" + RCCL gather + de-interleave" if multi else ""),
                       "frames_in_flight": n_lanes,
                       "tile_cost_estimate": "run in every frame (estimate_reused gives the figure with the library's reuse for an unchanged view)",
                       "pipeline": ("frame k runs on stream k %% %d with its own working set and output buffers; every frame builds its own grid; "
                                    "value = frames completed / wall time" % n_lanes) if n_lanes > 1 else "one frame at a time on one stream",
                       "partition": "interleaved 8-row bands, scene replicated" if world > 1 else "single GPU",
                       "triangles": cs["n_triangles"], "grid": cs["grid_n"], "cell_refs": cs["n_refs"]},
            "frame_ms": {"build": round(ms_build, 4), "render": round(ms_render, 4), "device_total": round(ms_build + ms_render, 4),
                         "note": "first build kernel to the start of the trace kernel, trace kernel, and their sum (the latency of a frame), "
                                 "as the frames of the timed region saw them, i.e. sharing the GPU with %d other frames" % (n_lanes - 1) if n_lanes > 1 else
                                 "first build kernel to the start of the trace kernel, trace kernel, their sum",
                         "alone": {"build": round(ms_build_alone, 4), "render": round(ms_render_alone, 4), "device_total": round(ms_build_alone + ms_render_alone, 4)},
                         "one_at_a_time": one_at_a_time,
                         "with_host_writeback": wb["ms"] if wb else None},
            "host_writeback": wb,
            "stress": stress,
            "render_only_mrays": round(rays_local / t_render_alone / 1e6, 2) if t_render_alone > 0 else None,
            "roofline": {
                "bound": "valu", "kernel": "k_trace",
                "achieved": round((f32 + f64) / t_launch / 1e12, 4), "peak": round((f32 + f64) / t_peak / 1e12, 2), "unit": "TFLOP/s",
                "frac": round(t_peak / t_launch, 5),
                "launch_ms": round(t_launch * 1e3, 4),
                "launch_ms_note": launch_note,
                "gpu_throughput_frac": round(t_peak / t_wall, 5),
                "gpu_throughput_note": "the same flops over the wall time per frame of the timed region (%d frames in flight, grid builds included): a figure of the whole "
                                       "GPU's throughput, not of a kernel launch" % n_lanes,
                "launch_in_flight": {"ms": round(ms_render, 4), "achieved": round((f32 + f64) / t_render / 1e12, 4), "frac": round(t_peak / t_render, 5),
                                     "note": "start to end of one k_trace launch in the timed region (the duration a profiler shows); with frames in flight it runs beside "
                                             "the kernels of %d other frames and is stretched by them" % (n_lanes - 1)},
                "alone": {"ms": round(ms_render_alone, 4), "achieved": round((f32 + f64) / t_render_alone / 1e12, 4), "frac": round(t_peak / t_render_alone, 5),
                          "note": "the same launch, same configuration, with the GPU to itself (frames one after the other on one stream, before the timed region)"},
                "one_at_a_time": ({"ms": one_at_a_time["render"], "achieved": round((f32 + f64) / (one_at_a_time["render"] * 1e-3) / 1e12, 4),
                                   "frac": round(t_peak / (one_at_a_time["render"] * 1e-3), 5),
                                   "note": "the launch as the library issues it for one frame at a time (every wavefront slot of the GPU, nothing beside it): "
                                           "the duration a kernel trace of `bench.py --in-flight 1` shows (profiles/*_one_at_a_time_kernel_stats.csv)"}
                                  if one_at_a_time and one_at_a_time["render"] > 0 else None),
                "traffic": traffic,
                "traffic_note": "HBM-side bytes of one k_trace launch from the L2 fabric counters (profiles/traffic.json: separate rocprofv3 --pmc passes, FETCH_SIZE doubled on gfx950)",
                "note": "flops of the triangle tests k_trace executes (%d; the reference's early-exit order needs %d, the rest are skipped by the per-run "
                        "box test), %d voxel steps and %d rays of rank 0's frame, over launch_ms (launch durations come from 100 MHz clock values the kernel writes at its start and end); "
                        "peak = the same flops at 157.3 TF fp32 + 78.6 TF fp64" % (tests, tests_ref, steps_, nrays),
            },
            "roofline_hbm": {
                "bound": "hbm", "achieved": round(hbm_bytes / t_launch / 1e9, 2), "peak": PEAK_HBM / 1e9, "unit": "GB/s",
                "frac": round(hbm_bytes / t_launch / PEAK_HBM, 6), "traffic": traffic,
                "traffic_rate": round(traffic / t_launch / 1e9, 2) if traffic else None,
                "note": "compulsory bytes (48 B per cell reference + cell table + RGB8 frame); the path is VALU-issue bound, not HBM bound",
            },
        }
        if not args.no_cpu:
            out["parity"], out["cpu_baseline"] = cpu_baseline(rt, scene, pkw, W, H, walk)
            if out.get("full_walk") is not None and out["parity"]["other_walk"]["gpu_walk"] == "full":
                out["full_walk"]["parity"] = out["parity"]["other_walk"]
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
