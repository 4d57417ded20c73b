#!/bin/bash
# round 6 (ON THE GPU BOX): the geometry alone (idle GPU), split per kernel, at N = 2048 / 4096 / 8192
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in cfg2 cfg4 cfg5; do
  rm -rf /tmp/gs_trace
  echo "== $cfg (tools/search_time.py $cfg under rocprofv3 --kernel-trace: one multi-stride launch + the single launches per step)"
  rocprofv3 --kernel-trace --stats -d /tmp/gs_trace -o t -- python $ROOT/tools/search_time.py $cfg 2>/dev/null | grep "geometry alone"
  python $ROOT/tools/pmc_query.py /tmp/gs_trace/t_results.db | grep -v "^no counters" | head -9
done
