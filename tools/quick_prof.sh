#!/bin/bash
# developer loop (run ON THE GPU BOX): GPU tests, bench with and without prefetch, rocprofv3 kernel summary
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd $ROOT
[ "${SKIP_TESTS:-0}" = 1 ] || python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py --steps 50 --warmup 10 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('prefetch   ', d['ms_per_step'], d['value'])"
python bench.py --steps 50 --warmup 10 --no-cpu --no-prefetch 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('no-prefetch', d['ms_per_step'], d['value'])"
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/qp_trace
rocprofv3 --kernel-trace --stats -d $OUT/qp_trace -o t -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-prefetch > $OUT/qp_trace.log 2>&1
python $ROOT/tools/pmc_query.py $OUT/qp_trace/t_results.db | head -14
rm -rf $OUT/qp_trace
