#!/bin/bash
# developer A/B (run ON THE GPU BOX): isolated kernel durations (bench --serial under rocprofv3) for each variants/lib_*.so
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in $ROOT/variants/lib_*.so; do
  echo "== $lib"
  rm -rf $OUT/ab_trace
  CONV3P_HIP_LIB=$lib rocprofv3 --kernel-trace --stats -d $OUT/ab_trace -o t -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-extra --serial > $OUT/ab_trace.log 2>&1
  python $ROOT/tools/pmc_query.py $OUT/ab_trace/t_results.db | grep "${AB_GREP:-conv3p}" | head -8 | cut -c1-110
done
rm -rf $OUT/ab_trace
