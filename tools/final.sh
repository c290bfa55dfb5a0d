#!/bin/bash
# end-of-round artefacts for profiles/: per-kernel stats (rocprofv3 --kernel-trace --stats) and the bench line of the same command,
# once as benchmarked (four frames in flight: launches overlap, so their durations are stretched) and once with one frame at
# a time (every kernel has the GPU to itself: the durations to compare kernels by)
mkdir -p gpurun_out/final; R=$PWD; export TMPDIR=/tmp
for w in "$@"; do
  for mode in "" one_at_a_time; do
    tag=${w}${mode:+_$mode}; flag=""; [ -n "$mode" ] && flag="--in-flight 1"
    cd /tmp
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/final/prof_$tag -o $w -- python3 $R/bench.py --no-cpu --no-extras --workload $w --steps 40 $flag > $R/gpurun_out/final/${ROUND:-r03}_final_${tag}_bench.json 2> $R/gpurun_out/final/$tag.err || { tail -5 $R/gpurun_out/final/$tag.err; exit 1; }
    cd $R
    f=$(ls gpurun_out/final/prof_$tag/*kernel_stats.csv gpurun_out/final/prof_$tag/*/*kernel_stats.csv 2>/dev/null | head -1)
    cp $f gpurun_out/final/${ROUND:-r03}_final_${tag}_kernel_stats.csv
    head -4 gpurun_out/final/${ROUND:-r03}_final_${tag}_kernel_stats.csv | cut -c1-90
  done
done
