"""Bursts of K frames between two synchronisations (the shape of a short timed region): ms per frame of a burst, fill and
drain of the pipeline included, against the rate of a long run (python tools/burst.py workload K depth...) (GPU)"""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
rt = importlib.import_module("ray-tracer_amd")
w, K = sys.argv[1], int(sys.argv[2])
scene, pkw, W, H, label = bench.build_workload(rt, w)
p = rt.default_params(W, H, **pkw)
scene.upload()
for depth in [int(x) for x in sys.argv[3:]]:
    scene.set_frames_in_flight(depth)
    hs = rt.create_streams(depth)
    outs = [torch.empty((H, W, 3), dtype=torch.uint8, device="cuda") for _ in hs]
    def run(m):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(m):
            scene.render_device(p, rebuild_grid=True, d_u8=outs[i % len(hs)].data_ptr(), stream=hs[i % len(hs)])
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / m * 1e3
    run(3 * depth)
    burst = sorted(run(K) for _ in range(9))[4]
    print(w, "depth", depth, "streams", len(hs), "bursts of %d: %.4f ms/frame (median of 9) | 200 frames: %.4f ms/frame" % (K, burst, run(200)), flush=True)
    rt.destroy_streams(hs)
