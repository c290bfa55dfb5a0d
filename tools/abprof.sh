#!/bin/bash
# per-kernel times under library variants: LIBS="a.so b.so" tools/abprof.sh workload kernel-name-substring
for l in $LIBS; do
  export RT_AMD_LIB=$PWD/ray-tracer_amd/$l
  ./tools/prof.sh $1 ab_$l > gpurun_out/abprof.txt 2>&1 || { tail -3 gpurun_out/abprof.txt; exit 1; }
  echo "$l: $(grep "$2" gpurun_out/abprof.txt | cut -d, -f1,4 | cut -c1-80)"
done
