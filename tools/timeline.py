"""developer: print the kernel timeline of the last steps of a rocprofv3 --kernel-trace database (start offsets in us).
usage: python tools/timeline.py <results.db> [n_kernels]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
q = "select name, start, end, %s from kernels order by start" % ("stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else "0"))
rows = list(db.execute(q))[-n:]
t0 = rows[0][1]
for name, s, e, st in rows:
    short = name.split("(")[0].replace("void conv3p::", "").replace("conv3p::", "")[:44]
    print("%-46s q%-3s start %9.1f  dur %7.1f  end %9.1f" % (short, st, (s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3))
