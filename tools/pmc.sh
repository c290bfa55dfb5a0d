#!/bin/bash
# usage: tools/pmc.sh <workload> <outdir> : SQ counter pass for k_trace
export TMPDIR=/tmp; R=$PWD; W=$1; O=$R/gpurun_out/$2; mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU --kernel-trace --output-format csv -d $O -o pmc -- python3 bench.py --no-cpu --no-extras --steps 2 --warmup 1 --workload $W > $O/bench.json 2> $O/err.log
python3 - <<PY
import csv,collections
rows=list(csv.DictReader(open("$O/pmc_counter_collection.csv")))
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in rows:
    k=r["Kernel_Name"][:28]
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k,v in agg.items():
    if "k_trace<false" in k or "classify" in k:
        d={c: v[c]/cnt[(k,c)] for c in v}
        print(k, {c: f"{x:.4g}" for c,x in d.items()})
        print("   lanes/VALU inst %.1f  wave_cycles/wave %.3g (x4 = cycles)" % (d["SQ_THREAD_CYCLES_VALU"]/d["SQ_ACTIVE_INST_VALU"], d["SQ_WAVE_CYCLES"]/d["SQ_WAVES"]))
PY
