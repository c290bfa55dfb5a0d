"""Rewrites the two end-of-round-2 tables and the closing lines of profiles/README.md from the r02_final_* files and
profiles/traffic.json (run from the repository root after copying gpurun_out/artefacts into profiles/)."""
import json, csv

def line(f):
    return json.loads([l for l in open(f) if l.startswith("{")][-1])

def ktrace_ms(f):
    for r in list(csv.reader(open(f)))[1:]:
        if r[0].replace("void ", "").startswith("k_trace<false"):
            return float(r[3]) / 1e6
    return 0.0

ws = ("nefertiti4k", "plane4k", "spot1080", "main512")
t1 = "| workload | Mrays/s | ms/frame (wall) | latency of a frame, ms | VALU roofline frac (wall) | frac (launch) |\n|---|---|---|---|---|---|\n"
t2 = "| workload | Mrays/s | ms/frame (wall) | build ms | trace ms (stamps) | k_trace ms (rocprof avg) | VALU roofline frac |\n|---|---|---|---|---|---|---|\n"
r1, r2 = "", ""
for w in ws:
    d = line(f"profiles/r02_final_{w}_bench.json")
    r1 += "| %s | %.0f | %.3f | %.3f | %.3f | %.3f |\n" % (w, d["value"], d["ms_per_step"], d["frame_ms"]["device_total"], d["roofline"]["frac"], d["roofline"]["launch_in_flight"]["frac"])
    d = line(f"profiles/r02_final_{w}_one_at_a_time_bench.json")
    r2 += "| %s | %.0f | %.3f | %.3f | %.3f | %.3f | %.3f |\n" % (w, d["value"], d["ms_per_step"], d["frame_ms"]["build"], d["frame_ms"]["render"],
                                                                  ktrace_ms(f"profiles/r02_final_{w}_one_at_a_time_kernel_stats.csv"), d["roofline"]["frac"])
s = open("profiles/README.md").read()
a = s.index(t1) + len(t1); b = s.index("\nOne frame at a time (")
s = s[:a] + r1 + s[b:]
a = s.index(t2, s.index("One frame at a time (")) + len(t2); b = s.index("\nIn the four-in-flight kernel lists")
s = s[:a] + r2 + s[b:]
d = line("profiles/r02_final_default_bench_with_cpu.json")
tr = json.load(open("profiles/traffic.json"))
a = s.index("Plain run without the profiler") if "Plain run without the profiler" in s else s.index("Plain runs without the profiler")
s = s[:a] + ("Plain run without the profiler (`r02_final_default_bench_with_cpu.json`): stand-in %.3f ms per frame (%.1f Grays/s) with four frames in\n"
             "flight, %.3f ms one at a time (build %.3f + trace %.3f), %.3f ms with the copy to pinned host memory, close camera %.2f ms; parity of the\n"
             "whole frame against the oracle: %d id mismatch(es), %d colour(s) off by more than 1/255.\n\n"
             "`k_trace` HBM-side traffic per launch (`traffic.json`; counter passes run `bench.py --in-flight 1`, the kernel as launched for one frame at a\n"
             "time, like the SQ counter passes): nefertiti4k %.2f GB, plane4k %.2f GB, spot1080 %.0f MB, main512 %.0f MB.\n"
             % (d["ms_per_step"], d["value"] / 1e3, d["frame_ms"]["one_at_a_time"]["ms_per_frame"], d["frame_ms"]["one_at_a_time"]["build"],
                d["frame_ms"]["one_at_a_time"]["render"], d["host_writeback"]["ms"], d["stress"]["ms_per_frame"], d["parity"]["id_mismatch"],
                d["parity"]["colour_gt_1_255"], tr["nefertiti4k"]["bytes_per_launch"] / 1e9, tr["plane4k"]["bytes_per_launch"] / 1e9,
                tr["spot1080"]["bytes_per_launch"] / 1e6, tr["main512"]["bytes_per_launch"] / 1e6))
open("profiles/README.md", "w").write(s)
print(r1); print(r2)
