#!/bin/bash
# random views against the oracle: the reference's scene, the scene with crowded voxels (run culling, pre-filter), frames in flight
mkdir -p gpurun_out
{
  echo "== tools/fuzz.py 40 31 (spot + blub)"; timeout -k 10 500 python tools/fuzz.py 40 31 2>&1 | tail -4 || exit 1
  echo "== tools/fuzz.py 30 32 3 (spot + blub subdivided 3x: crowded voxels)"; timeout -k 10 700 python tools/fuzz.py 30 32 3 2>&1 | tail -4 || exit 1
  echo "== tools/fuzz_flight.py 24 33 3 (four frames in flight, random shares)"; timeout -k 10 700 python tools/fuzz_flight.py 24 33 3 2>&1 | tail -4 || exit 1
} > gpurun_out/fuzz_r03.txt 2>&1
cat gpurun_out/fuzz_r03.txt
