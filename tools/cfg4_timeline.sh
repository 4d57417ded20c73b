#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/c4_trace
rocprofv3 --kernel-trace -d $OUT/c4_trace -o t -- python $ROOT/tools/cfg4_step.py $@ > $OUT/c4_trace.log 2>&1
tail -1 $OUT/c4_trace.log
python $ROOT/tools/timeline.py $OUT/c4_trace/t_results.db 36
rm -rf $OUT/c4_trace
