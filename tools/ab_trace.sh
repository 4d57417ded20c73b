#!/bin/bash
# developer A/B (ON THE GPU BOX): per-kernel average durations of the headline bench (overlapped) and of --serial
# (every kernel alone) for each library given:  tools/ab_trace.sh devlibs/lib_a.so devlibs/lib_b.so ...
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  for mode in "" "--serial"; do
    rm -rf $OUT/abt
    CONV3P_HIP_LIB=$ROOT/$lib rocprofv3 --kernel-trace --stats -d $OUT/abt -o t -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-extra $mode > $OUT/abt.log 2>&1
    echo "== $lib $mode"
    python $ROOT/tools/pmc_query.py $OUT/abt/t_results.db | head -13 | tail -12
  done
done
rm -rf $OUT/abt
