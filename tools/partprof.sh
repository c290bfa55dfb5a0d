#!/bin/bash
# per-kernel times of one rank's share of a frame: tools/partprof.sh <workload> <parts> <tag>
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/pprof_$3 -o $3 --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/partbench.py $1 $2 > $GRAFT_REPO_ROOT/gpurun_out/pprof_$3.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/pprof_$3.log; exit 1; }
f=$(ls $GRAFT_REPO_ROOT/gpurun_out/pprof_$3/*/*kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/pprof_$3/*kernel_stats.csv 2>/dev/null | head -1)
cut -d, -f1-5 $f | cut -c1-120
grep part $GRAFT_REPO_ROOT/gpurun_out/pprof_$3.log
