#!/bin/bash
# developer (ON THE GPU BOX): per-kernel durations of tools/shape_time.py <args> under rocprofv3 --kernel-trace
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/shp_trace
rocprofv3 --kernel-trace --stats -d $OUT/shp_trace -o t -- python $ROOT/tools/shape_time.py "$@" > $OUT/shp_trace.log 2>&1
tail -1 $OUT/shp_trace.log
python $ROOT/tools/pmc_query.py $OUT/shp_trace/t_results.db | head -12
rm -rf $OUT/shp_trace
