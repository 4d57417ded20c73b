#!/bin/bash
# developer (ON THE GPU BOX): per-kernel durations of the headline mode (side-stream overlap)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/os_trace
rocprofv3 --kernel-trace --stats -d $OUT/os_trace -o t -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-extra $@ > $OUT/os_trace.log 2>&1
python $ROOT/tools/pmc_query.py $OUT/os_trace/t_results.db | head -14
rm -rf $OUT/os_trace
