#!/bin/bash
# developer (ON THE GPU BOX): geometry time alone for every devlibs/lib_*.so (timing builds with phases removed) and, with
# PMC=1, the vector / scalar / LDS instruction counts of the multi-stride search launch
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for lib in $ROOT/devlibs/lib_*.so; do
  export CONV3P_HIP_LIB=$lib
  echo "$(basename $lib): $(python $ROOT/tools/search_time.py ${CFG:-cfg2} 2>/dev/null | tail -1)"
  if [ "${PMC:-0}" = 1 ]; then
    rm -rf $OUT/sa_pmc
    rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $OUT/sa_pmc -o p -- python $ROOT/tools/search_time.py ${CFG:-cfg2} > $OUT/sa_pmc.log 2>&1
    python - <<PY
import sqlite3
db = sqlite3.connect("$OUT/sa_pmc/p_results.db")
cur = db.execute("select * from counters_collection limit 1"); cols = [d[0] for d in cur.description]
ki = "kernel_name" if "kernel_name" in cols else "name"
# multi launches have 4x the waves of single launches: split by SQ_WAVES
rows = list(db.execute("select dispatch_id, %s, counter_name, value from counters_collection" % ki))
by = {}
for d, n, c, v in rows:
    if "search" in n: by.setdefault(d, {})[c] = v
groups = {}
for d, cs in by.items():
    groups.setdefault(int(cs.get("SQ_WAVES", 0)), []).append(cs)
for w, lst in sorted(groups.items()):
    print("   waves %6d launches %3d  VALU %.4g  SALU %.4g  LDS %.4g" % (w, len(lst), sum(c["SQ_INSTS_VALU"] for c in lst) / len(lst), sum(c["SQ_INSTS_SALU"] for c in lst) / len(lst), sum(c["SQ_INSTS_LDS"] for c in lst) / len(lst)))
PY
  fi
done
rm -rf $OUT/sa_pmc
