"""Counters of one counting frame of a workload (steps, tests, trips, phase cycles, loop counts): python tools/counts.py [workload] (GPU)"""
import ctypes as C, importlib, json, sys, os
sys.path.insert(0, os.getcwd())
import torch, bench
rt = importlib.import_module("ray-tracer_amd")
LOOPS = ["rounds_at_voxel", "advance", "step_iter", "cull_a_rounds", "cull_a_iter", "cull_a_lanes", "cull_b_voxels", "cull_b_rays", "itemgen_calls",
         "itemgen_iter", "itemgen_voxels", "items", "items_with_rays", "trips", "trips_cand", "trips_pass", "pass_lanes", "walks", "filter_kept"]
w = sys.argv[1] if len(sys.argv) > 1 else "nefertiti4k"
scene, pkw, W, H, label = bench.build_workload(rt, w)
scene.upload()
p = rt.default_params(W, H, **pkw)
buf = torch.empty((H, W, 3), dtype=torch.uint8, device="cuda")
st = torch.cuda.current_stream().cuda_stream
scene.render_device(p, rebuild_grid=True, d_u8=buf.data_ptr(), stream=st)
scene.set_counting(True)
N = len(LOOPS)
out = (C.c_uint64 * (2 * N))()
rt.lib().rt_loop_counts(out, 2 * N, 1)
scene.render_device(p, rebuild_grid=False, d_u8=buf.data_ptr(), stream=st)
torch.cuda.synchronize()
cs = scene.last_stats()
rt.lib().rt_loop_counts(out, 2 * N, 1)
for k in sorted(cs): print(k, cs[k])
loops = {"primary": {n: int(out[i]) for i, n in enumerate(LOOPS)}, "shadow": {n: int(out[N + i]) for i, n in enumerate(LOOPS)}}
for n in LOOPS: print("loop %-16s primary %10d shadow %10d" % (n, loops["primary"][n], loops["shadow"][n]))
json.dump({"workload": label, "counters": {k: (v if not hasattr(v, "tolist") else v.tolist()) for k, v in cs.items()}, "loops": loops}, open("gpurun_out/counts_%s.json" % w, "w"), indent=1, default=str)
