#!/bin/bash
# developer (ON THE GPU BOX): kernel timeline of the last steps of the headline bench (both streams)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/tl_trace
rocprofv3 --kernel-trace -d $OUT/tl_trace -o t -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-extra > $OUT/tl_trace.log 2>&1
python $ROOT/tools/timeline.py $OUT/tl_trace/t_results.db ${1:-90} | head -60
rm -rf $OUT/tl_trace
