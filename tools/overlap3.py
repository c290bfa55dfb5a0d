"""Frames in flight: ms per frame of one scene rendered round-robin on `depth` streams, each frame with its own grid build
(python tools/overlap3.py workload parts depth...) (GPU)"""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
rt = importlib.import_module("ray-tracer_amd")
w, parts = sys.argv[1], int(sys.argv[2])
scene, pkw, W, H, label = bench.build_workload(rt, w)
p = rt.default_params(W, H, part_index=0, part_count=parts, **pkw)
scene.upload()
for depth in [int(x) for x in sys.argv[3:]]:
    scene.set_frames_in_flight(depth)
    hs = rt.create_streams(depth)
    outs = [torch.empty((rt.part_rows(p), W, 3), dtype=torch.uint8, device="cuda") for _ in hs]
    def run(m):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(m):
            scene.render_device(p, rebuild_grid=os.environ.get("RT_NO_REBUILD", "") != "1", d_u8=outs[i % len(hs)].data_ptr(), stream=hs[i % len(hs)])
        t_host = (time.perf_counter() - t0) / m * 1e3
        torch.cuda.synchronize(); run.host = t_host; return (time.perf_counter() - t0) / m * 1e3
    run(3 * depth)
    t = run(96)
    lat = [scene.stats_at(k)["ms_total"] for k in range(32)]
    print(w, "parts", parts, "depth", depth, "streams", len(hs), "%.4f ms/frame, frame latency %.3f ms, host enqueue %.4f ms/frame" % (t, sum(lat) / len(lat), run.host), flush=True)
    rt.destroy_streams(hs)
