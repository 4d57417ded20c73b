#!/bin/bash
# developer A/B (run ON THE GPU BOX): headline bench (default and --serial) for each devlibs/lib_*.so, 3 repetitions
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT
for rep in 1 2 3; do
for lib in devlibs/lib_*.so; do
  for mode in "" "--serial"; do
    CONV3P_HIP_LIB=$ROOT/$lib python bench.py --steps 50 --warmup 10 --no-cpu --no-extra $mode 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', '$mode', d['ms_per_step'], d['value'])"
  done
done
done
