#!/bin/bash
# one rank's share of a frame under several unit-size settings: SHIFTS="a b c" tools/partsweep.sh workload N...
w=$1; shift
for sh in $SHIFTS; do
  echo "== RT_SHIFTS=$sh"
  if [ "$sh" = "default" ]; then unset RT_SHIFTS; else export RT_SHIFTS=$sh; fi
  timeout -k 10 200 python tools/partbench.py $w "$@" 2>/dev/null | grep "part 0" || exit 1
done
