#!/bin/bash
# per-kernel times of one workload: tools/prof.sh <workload> <tag>
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_$2 -o $2 --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-extras --workload $1 --steps 20 > $GRAFT_REPO_ROOT/gpurun_out/prof_$2.log 2>&1 || { tail -5 $GRAFT_REPO_ROOT/gpurun_out/prof_$2.log; exit 1; }
f=$(ls $GRAFT_REPO_ROOT/gpurun_out/prof_$2/*/*kernel_stats.csv $GRAFT_REPO_ROOT/gpurun_out/prof_$2/*kernel_stats.csv 2>/dev/null | head -1)
cut -d, -f1-5 $f | cut -c1-110
