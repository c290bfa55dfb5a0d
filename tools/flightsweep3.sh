#!/bin/bash
# unit sizes x blocks per CU with four frames in flight
for w in ${WL:-nefertiti4k plane4k spot1080 main512 nefertiti4k_close}; do
  for sh in adapt 0000000000000000 0000000000001111 0000000011111111 0000000000002222 0000000011112222 1111000011112222; do
    for b in 1 2 4; do
      if [ $sh = adapt ]; then e="RT_X=0"; else e="RT_SHIFTS=$sh"; fi
      r=$(env $e RT_BLOCKS_PER_CU=$b timeout -k 10 200 python tools/overlap3.py $w ${P:-1} 4 2>/dev/null | grep depth) || exit 1
      echo "$w $sh bpc$b $(echo $r | sed 's/.*streams 4 //')"
    done
  done
done
