"""FETCH_SIZE / WRITE_SIZE (two separate rocprofv3 --pmc passes) -> per-kernel HBM bytes per launch.

rocprofv3 reports both counters in KiB.  MI355X_MICROARCH.md (HBM section): on gfx950 FETCH_SIZE counts a wide
(16 B/lane) coalesced streaming read at exactly half its bytes; other access widths are uncalibrated.  The
conv3p kernels mix 16-B record loads with 4-B gathers, so both the raw figure and the doubled-read figure are
stored: traffic = write + 2*fetch is an UPPER bound, write + fetch a lower bound."""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pointwise_amd.build import source_hash


def per_kernel(db, counter):
    c = sqlite3.connect(db)
    cur = c.execute("select * from counters_collection limit 1")
    cols = [d[0] for d in cur.description]
    ki = "kernel_name" if "kernel_name" in cols else "name"
    out = {}
    q = "select %s, avg(value), count(*) from counters_collection where counter_name = ? group by %s" % (ki, ki)
    for name, avg, n in c.execute(q, (counter,)):
        key = name.split("(")[0].replace("void ", "").replace("conv3p::", "").split("<")[0]
        key = {"backward_sparse_kernel": "backward_kernel", "search_fused_kernel": "search_kernel",
               "search_multi_kernel": "search_kernel"}.get(key, key)   # bench.py's kinds
        a = out.setdefault(key, [0.0, 0])
        a[0] += avg * n
        a[1] += n
    return {k: v[0] / v[1] for k, v in out.items()}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
res = {}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0) * 1024.0, write.get(k, 0.0) * 1024.0
    res[k] = {"fetch_bytes_raw": round(f), "write_bytes": round(w), "hbm_bytes_lower": round(f + w),
              "hbm_bytes_upper_fetch_x2": round(2 * f + w)}
res["_csrc_sha"] = source_hash()      # bench.py drops these counters when the library's sources have changed since
print(json.dumps(res, indent=1))
