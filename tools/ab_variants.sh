#!/bin/bash
# developer (ON THE GPU BOX): isolated kernel durations for every variants/lib_*.so (bench.py --serial under rocprofv3)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
for lib in $ROOT/variants/lib_*.so; do
  rm -rf /tmp/tr
  CONV3P_HIP_LIB=$lib rocprofv3 --kernel-trace --stats -d /tmp/tr -o t -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-extra --serial > /tmp/tr.log 2>&1
  echo "== $(basename $lib)"; python $ROOT/tools/pmc_query.py /tmp/tr/t_results.db | grep "${1:-backward}"
done
