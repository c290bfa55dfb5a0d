#!/bin/bash
# HBM-side bytes of k_trace per launch from the L2 fabric counters, one rocprofv3 pass per counter:
#   tools/traffic.sh <workload>  ->  gpurun_out/traffic_<workload>.json
export TMPDIR=/tmp; R=$PWD; W=$1; O=$R/gpurun_out/traffic_$W; mkdir -p $O
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -s KILL 120 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/$C -o pmc -- python3 bench.py --no-cpu --no-extras --in-flight 1 --steps 3 --warmup 1 --workload $W > $O/$C.json 2> $O/$C.err || { tail -3 $O/$C.err; exit 1; }
done
python3 - <<PY
import csv, json, glob
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("$O/%s/**/pmc_counter_collection.csv" % c, recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_trace<false" in r["Kernel_Name"] and r["Counter_Name"] == c]
    out[c + "_KB_per_launch"] = sum(v) / len(v)
    out[c + "_launches"] = len(v)
# gfx950: FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes -> doubled; WRITE_SIZE is exact
out["bytes_per_launch"] = out["FETCH_SIZE_KB_per_launch"] * 1024 * 2 + out["WRITE_SIZE_KB_per_launch"] * 1024
out["kernel"] = "k_trace"; out["workload"] = "$W"
json.dump(out, open("$R/gpurun_out/traffic_$W.json", "w"), indent=1)
print(json.dumps(out))
PY
