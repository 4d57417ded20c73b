#!/bin/bash
# developer (run ON THE GPU BOX): rocprofv3 kernel summary of tools/shape_time.py for one layer shape
#   bash tools/shape_prof.sh 36 41 16 4096 room
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/sp_trace
rocprofv3 --kernel-trace --stats -d $OUT/sp_trace -o t -- python $ROOT/tools/shape_time.py "$@" > $OUT/sp.log 2>&1
tail -1 $OUT/sp.log
python $ROOT/tools/pmc_query.py $OUT/sp_trace/t_results.db | head -${SP_LINES:-14} | cut -c1-120
rm -rf $OUT/sp_trace
