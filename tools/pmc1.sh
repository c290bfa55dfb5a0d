#!/bin/bash
# one SQ counter pass of the trace kernel (instruction counts): tools/pmc1.sh <workload>
export TMPDIR=/tmp; R=$PWD; W=${1:-nefertiti4k}; O=$R/gpurun_out/pmc1_$W; rm -rf $O; mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $O -o pmc -- python3 bench.py --no-cpu --no-extras --in-flight 1 --steps 2 --warmup 1 --workload $W > $O/run.json 2> $O/run.err || { tail -3 $O/run.err; exit 1; }
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$O/**/pmc_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace<false" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c in sorted(agg):
    v = agg[c]; print("%-24s %14.6g  (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
