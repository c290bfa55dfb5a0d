#!/bin/bash
# Everything under profiles/ for one round, run on the GPU box from the repository root: ROUND=r02 tools/round_artefacts.sh
# kernel stats + bench lines of the four workloads, the default bench line with the CPU leg, HBM traffic of k_trace
# (FETCH_SIZE / WRITE_SIZE passes) and the SQ counter passes. Results land in gpurun_out/artefacts/ (copy them to profiles/).
R=$PWD; export ROUND=${ROUND:-r03}; A=$R/gpurun_out/artefacts; mkdir -p $A
./tools/final.sh nefertiti4k plane4k spot1080 main512 || exit 1
cp gpurun_out/final/${ROUND}_final_* $A/
python3 - <<PY > $A/traffic.json
import json, subprocess, sys
out = {}
for w in ("nefertiti4k", "plane4k", "spot1080", "main512"):
    r = subprocess.run(["./tools/traffic.sh", w], capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if not line:
        sys.stderr.write(r.stdout + r.stderr); continue
    d = json.loads(line[-1])
    out[w] = {"bytes_per_launch": int(d["bytes_per_launch"]), "kernel": "k_trace", "FETCH_SIZE_KB": round(d["FETCH_SIZE_KB_per_launch"], 1),
              "WRITE_SIZE_KB": round(d["WRITE_SIZE_KB_per_launch"], 1),
              "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (tools/traffic.sh: bench.py --in-flight 1, the kernel as launched for one frame at a time), mean over %d launches; bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 (gfx950 tallies the 128-byte requests of 16-byte-per-lane reads at 64 bytes; Infinity-Cache hits are included)" % d["FETCH_SIZE_launches"]}
print(json.dumps(out, indent=1))
PY
# the default bench line reads the traffic figures of this same round
cp $A/traffic.json profiles/traffic.json
timeout -k 10 400 python -u bench.py > $A/${ROUND}_final_default_bench_with_cpu.json 2> $A/default.err || { tail -5 $A/default.err; exit 1; }
for w in nefertiti4k plane4k spot1080; do
  ./tools/pmc2.sh $w ${ROUND}_$w > $A/${ROUND}_pmc_${w}_summary.txt || exit 1
  cp gpurun_out/pmc_${ROUND}_$w/summary.csv $A/${ROUND}_pmc_${w}_k_trace.csv
  for k in 1 2 3 4; do cp gpurun_out/pmc_${ROUND}_$w/pass$k/pmc_counter_collection.csv $A/${ROUND}_pmc_${w}_pass$k.csv; done
done
ls $A
