#!/bin/bash
# round 6 (ON THE GPU BOX): cfg4 step with / without the fused stack launches, with and without prefetch
for mode in fused perlayer; do
  f=""; [ $mode = fused ] && f="--fused"
  for a in "" "--no-prefetch" "--sparse" "--sparse --no-prefetch"; do
    echo -n "$mode $a: "; timeout 200 python tools/cfg4_step.py $f $a 2>/dev/null | tail -1
    echo -n "$mode $a: "; timeout 200 python tools/cfg4_step.py $f $a 2>/dev/null | tail -1
  done
done
