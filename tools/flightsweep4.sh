#!/bin/bash
run() { echo -n "$1: "; shift; r=$(env "$@" timeout -k 10 200 python tools/overlap3.py $W $P 4 2>/dev/null | grep depth) || exit 1; echo "$r" | sed 's/.*streams 4 //'; }
for wp in "nefertiti4k 1" "nefertiti4k 2" "nefertiti4k 4" "nefertiti4k 8" "plane4k 1" "plane4k 8" "spot1080 1" "main512 1" "nefertiti4k_close 1"; do
  set -- $wp; export W=$1 P=$2
  run "$W/$P default" RT_X=0
  run "$W/$P bpc1" RT_BLOCKS_PER_CU=1
  run "$W/$P bpc2" RT_BLOCKS_PER_CU=2
  run "$W/$P bpc4" RT_BLOCKS_PER_CU=4
done
