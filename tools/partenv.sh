#!/bin/bash
# one rank's share of a frame under several environment settings: tools/partenv.sh workload N "A=1" "B=2" ...
w=$1; n=$2; shift; shift
for cfg in "$@"; do
  echo "$cfg"; env $cfg timeout -k 10 200 python tools/partbench.py $w $n 2>/dev/null | grep "part 0"
done
