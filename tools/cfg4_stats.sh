#!/bin/bash
# developer (ON THE GPU BOX): per-kernel durations of the cfg4 stack (tools/stack_time.py cfg4)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/c4
rocprofv3 --kernel-trace --stats -d /tmp/c4 -o t -- python $ROOT/tools/stack_time.py cfg4 > /tmp/c4.log 2>&1
tail -1 /tmp/c4.log
python $ROOT/tools/pmc_query.py /tmp/c4/t_results.db | head -14
