#!/bin/bash
# developer (ON THE GPU BOX): A/B of the populated-rows backward with tiles over the capacity split by centres (halves) or by taps (base)
cd "$(dirname "$0")/.."
for lib in base halves; do
  export CONV3P_HIP_LIB=$PWD/devlibs/lib_$lib.so
  for s in 1 2; do python tools/shape_time.py 36 13 16 4096 room $s 2>&1 | tail -1; done
  python tools/shape_time.py 36 13 16 8192 room 1 2>&1 | tail -1
  python tools/cfg4_step.py 2>&1 | tail -1
  python tools/cfg4_step.py --no-prefetch 2>&1 | tail -1
done
