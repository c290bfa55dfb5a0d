"""Rewrites the end-of-round table in profiles/README.md from the r01_final_* files."""
import json, csv, re
rows = []
for w in ("nefertiti4k", "plane4k", "spot1080", "main512"):
    d = json.load(open(f"profiles/r01_final_{w}_bench.json"))
    ks = {r[0].split('(')[0].replace('void ', ''): float(r[3]) / 1e6 for r in list(csv.reader(open(f"profiles/r01_final_{w}_kernel_stats.csv")))[1:]}
    rows.append((w, d["value"], d["ms_per_step"], d["frame_ms"]["build"], d["frame_ms"]["render"], next((v for k, v in ks.items() if k.startswith("k_trace<false")), 0), d["roofline"]["frac"]))
s = open("profiles/README.md").read()
head = "| workload | Mrays/s | ms/frame (wall) | build ms | trace ms (events) | k_trace ms (rocprof avg) | VALU roofline frac |\n|---|---|---|---|---|---|---|\n"
a = s.index(head) + len(head)
b = s.index("\nThe rocprof average")
s = s[:a] + "".join("| %s | %.0f | %.3f | %.3f | %.3f | %.3f | %.3f |\n" % r for r in rows) + s[b:]
open("profiles/README.md", "w").write(s)
print("".join("| %s | %.0f | %.3f | %.3f | %.3f | %.3f | %.3f |\n" % r for r in rows))
