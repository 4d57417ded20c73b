#!/bin/bash
# developer (run ON THE GPU BOX): duration of the search kernel per stride (bench --serial launches strides 1..4 in order)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf $OUT/st_trace
rocprofv3 --kernel-trace --stats -d $OUT/st_trace -o t -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu --no-extra --serial > $OUT/st_trace.log 2>&1
python - <<PY
import sqlite3
db = sqlite3.connect("$OUT/st_trace/t_results.db")
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
rows = list(db.execute("select name, start, duration from kernels where name like '%search_kernel%' order by start"))
rows = rows[-80:]                      # the last 20 steps: 4 launches each, strides 1, 2, 3, 4
for k in range(4):
    d = [r[2] for i, r in enumerate(rows) if i % 4 == k]
    print("launch %d of a step: avg %.1f us over %d" % (k, sum(d) / len(d) / 1e3, len(d)))
PY
rm -rf $OUT/st_trace
