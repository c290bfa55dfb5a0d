#!/bin/bash
# SQ counter passes for the trace kernel, one rocprofv3 run per pass (at most 8 SQ counters each):
#   tools/pmc2.sh <workload> <tag>   ->  gpurun_out/pmc_<tag>/pass<k>/..., summary on stdout and in gpurun_out/pmc_<tag>/summary.csv
export TMPDIR=/tmp; R=$PWD; W=$1; O=$R/gpurun_out/pmc_$2; mkdir -p $O
P[1]="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_INSTS_SALU"
P[2]="SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32"
P[3]="SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM"
P[4]="SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_TRANS_F32"
for k in 1 2 3 4; do
  timeout -s KILL 120 rocprofv3 --pmc ${P[$k]} --kernel-trace --output-format csv -d $O/pass$k -o pmc -- python3 bench.py --no-cpu --no-extras --in-flight 1 --steps 2 --warmup 1 --workload $W > $O/pass$k.json 2> $O/pass$k.err || { tail -3 $O/pass$k.err; exit 1; }
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$O/pass*/**/pmc_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_trace<false" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
with open("$O/summary.csv", "w") as out:
    out.write("counter,mean_per_launch,launches\n")
    for c in sorted(agg):
        v = agg[c]; out.write("%s,%.6g,%d\n" % (c, sum(v) / len(v), len(v)))
        print("%-28s %14.6g  (%d launches)" % (c, sum(v) / len(v), len(v)))
PY
