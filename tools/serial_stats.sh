#!/bin/bash
# developer (ON THE GPU BOX): per-kernel durations with every kernel alone on the GPU (bench.py --serial)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/ss_trace
rocprofv3 --kernel-trace --stats -d $OUT/ss_trace -o t -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-extra --serial $@ > $OUT/ss_trace.log 2>&1
python $ROOT/tools/pmc_query.py $OUT/ss_trace/t_results.db | head -14
rm -rf $OUT/ss_trace
