#!/bin/bash
run() { echo "== $1"; shift; env "$@" timeout -k 10 200 python tools/overlap3.py ${W:-nefertiti4k} ${P:-1} ${D:-4} 2>/dev/null | grep depth || exit 1; }
for w in nefertiti4k plane4k spot1080 main512 nefertiti4k_close; do
  export W=$w
  run "$w default" RT_X=0
  run "$w bpc1" RT_BLOCKS_PER_CU=1
  run "$w bpc2" RT_BLOCKS_PER_CU=2
done
W=nefertiti4k D="2 3" run "bpc2 depth 2 3" RT_BLOCKS_PER_CU=2
W=nefertiti4k D="8" run "bpc1 depth 8 (4 queues)" RT_BLOCKS_PER_CU=1
W=nefertiti4k D="4" run "bpc2 shifts0" RT_BLOCKS_PER_CU=2 RT_SHIFTS=0000000000000000
W=nefertiti4k D="4" run "bpc2 shifts1" RT_BLOCKS_PER_CU=2 RT_SHIFTS=1111111111111111
W=nefertiti4k D="4" run "bpc2 sched 0.5:2" RT_BLOCKS_PER_CU=2 RT_SCHED=0.5:2
