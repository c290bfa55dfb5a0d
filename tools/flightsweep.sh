#!/bin/bash
# Scheduling knobs of the trace kernel with four frames in flight (the tail of a frame no longer costs wall time there)
run() { echo "== $1"; shift; env "$@" timeout -k 10 200 python tools/overlap3.py ${W:-nefertiti4k} ${P:-1} 1 4 2>/dev/null | grep depth || exit 1; }
run default RT_X=0
run shifts0 RT_SHIFTS=0000000000000000
run shifts1 RT_SHIFTS=1111111111111111
run shifts2 RT_SHIFTS=2222222222222222
run shifts_exp0 RT_SHIFTS=2222000000002222
run sched_2_8 RT_SCHED=2:8
run sched_05_2 RT_SCHED=0.5:2
run bpc2 RT_BLOCKS_PER_CU=2
run bpc3 RT_BLOCKS_PER_CU=3
run cull32 RT_CULL_MIN=32
run cull128 RT_CULL_MIN=128
