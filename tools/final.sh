#!/bin/bash
# end-of-round artefacts for profiles/: per-kernel stats (rocprofv3 --kernel-trace --stats) and the bench line of the same command
mkdir -p gpurun_out/final; R=$PWD; export TMPDIR=/tmp
for w in "$@"; do
  cd /tmp
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof_$w -o $w -- python3 $R/bench.py --no-cpu --no-extras --workload $w --steps 20 > $R/gpurun_out/final/${ROUND:-r02}_final_${w}_bench.json 2> $R/gpurun_out/final/$w.err || { tail -5 $R/gpurun_out/final/$w.err; exit 1; }
  cd $R
  f=$(ls gpurun_out/final/prof_$w/*kernel_stats.csv gpurun_out/final/prof_$w/*/*kernel_stats.csv 2>/dev/null | head -1)
  cp $f gpurun_out/final/${ROUND:-r02}_final_${w}_kernel_stats.csv
  head -4 gpurun_out/final/${ROUND:-r02}_final_${w}_kernel_stats.csv | cut -c1-90
done
