#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/ob_trace
rocprofv3 --kernel-trace -d $OUT/ob_trace -o t -- python $ROOT/tools/op_boundary.py > $OUT/ob_trace.log 2>&1
tail -1 $OUT/ob_trace.log
python $ROOT/tools/timeline.py $OUT/ob_trace/t_results.db ${1:-70}
rm -rf $OUT/ob_trace
