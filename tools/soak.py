"""Soak run: N frames of one workload with frames in flight, every frame with its own grid build; every lane's buffer is
compared with the first frame's bytes every `check` frames (python tools/soak.py workload N [parts]) (GPU)"""
import importlib, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
rt = importlib.import_module("ray-tracer_amd")
w, N = sys.argv[1], int(sys.argv[2])
parts = int(sys.argv[3]) if len(sys.argv) > 3 else 1
scene, pkw, W, H, label = bench.build_workload(rt, w)
p = rt.default_params(W, H, part_index=0, part_count=parts, **pkw)
scene.upload()
scene.set_frames_in_flight(4)
hs = rt.create_streams(4)
outs = [torch.zeros((rt.part_rows(p), W, 3), dtype=torch.uint8, device="cuda") for _ in hs]
ref = torch.zeros_like(outs[0])
scene.render_device(p, rebuild_grid=True, d_u8=ref.data_ptr(), stream=hs[0])
torch.cuda.synchronize()
bad = 0; check = 200; t0 = time.perf_counter()
for i in range(N):
    scene.render_device(p, rebuild_grid=True, d_u8=outs[i % len(hs)].data_ptr(), stream=hs[i % len(hs)])
    if (i + 1) % check == 0:
        torch.cuda.synchronize()
        bad += sum(0 if torch.equal(o, ref) else 1 for o in outs)
        for o in outs: o.zero_()
        print("frame", i + 1, "buffers that differ so far:", bad, "%.3f ms/frame" % ((time.perf_counter() - t0) / (i + 1) * 1e3), flush=True)
torch.cuda.synchronize()
print(w, "parts", parts, "frames", N, "buffers that differed:", bad)
sys.exit(1 if bad else 0)
