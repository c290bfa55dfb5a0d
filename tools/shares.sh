#!/bin/bash
# one rank's share (part 0 of 1, 2, 4, 8) of the frames of configs 3, 4 and 5 on one GPU, one frame at a time and four in flight
mkdir -p gpurun_out
for w in "$@"; do
  for parts in 1 2 4 8; do
    echo "== $w parts $parts"
    timeout -k 10 200 python tools/overlap3.py $w $parts 1 4 2>/dev/null | grep "ms/frame" || exit 1
  done
done
