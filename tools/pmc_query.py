"""Summarise a rocprofv3 rocpd sqlite database: per-kernel average duration and PMC counter sums.
usage: python tools/pmc_query.py <results.db> [substring-of-kernel-name]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cur = db.execute("select name, count(*), avg(duration), sum(duration) from kernels group by name order by sum(duration) desc")
print("%-60s %6s %12s %12s" % ("kernel", "calls", "avg_us", "total_ms"))
for name, n, avg, tot in cur:
    if flt in name:
        print("%-60s %6d %12.2f %12.3f" % (name[:60], n, avg / 1e3, tot / 1e6))
try:
    cur = db.execute("select * from counters_collection limit 1")
    cols = [d[0] for d in cur.description]
    ki = "kernel_name" if "kernel_name" in cols else "name"
    q = "select %s, counter_name, count(*), avg(value) from counters_collection group by %s, counter_name" % (ki, ki)
    print()
    for name, cname, n, avg in db.execute(q):
        if flt in name:
            print("%-50s %-28s n=%-5d avg=%.4g" % (name[:50], cname, n, avg))
except Exception as e:
    print("no counters:", e)
