#!/bin/bash
# developer loop (run ON THE GPU BOX): per-kernel durations with every kernel alone on the GPU -- the cfg2 bench step
# with --serial, and the cfg4 stack (tools/stack_time.py) -- from rocprofv3 kernel traces
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/ss_trace
rocprofv3 --kernel-trace --stats -d $OUT/ss_trace -o t -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-extra --serial > $OUT/ss_trace.log 2>&1
python $ROOT/tools/pmc_query.py $OUT/ss_trace/t_results.db | head -12
rm -rf $OUT/ss_trace
rocprofv3 --kernel-trace --stats -d $OUT/ss_trace -o t -- python $ROOT/tools/stack_time.py > $OUT/ss_stack.log 2>&1
tail -3 $OUT/ss_stack.log
python $ROOT/tools/pmc_query.py $OUT/ss_trace/t_results.db | head -16
rm -rf $OUT/ss_trace
