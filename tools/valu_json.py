"""SQ_INSTS_VALU (+ the other SQ counters of the same rocprofv3 --pmc pass) -> per-kernel averages per launch, keyed
like bench.py's HIP-event kinds (kernel base name without template arguments).
usage: python tools/valu_json.py <pmc_SQ_WAVES results.db> > profiles/valu_latest.json"""
import json
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.execute("select * from counters_collection limit 1")
cols = [d[0] for d in cur.description]
ki = "kernel_name" if "kernel_name" in cols else "name"
acc = {}
for name, cname, n, avg in db.execute(
        "select %s, counter_name, count(*), avg(value) from counters_collection group by %s, counter_name" % (ki, ki)):
    if "conv3p::" not in name:
        continue
    key = name.split("(")[0].replace("void ", "").replace("conv3p::", "").split("<")[0]
    key = {"prep_sort_kernel": "prep_kernel",
           "reduce_multi_kernel": "reduce_partials_kernel", "backward_sparse_kernel": "backward_kernel"}.get(key, key)
    a = acc.setdefault(key, {}).setdefault(cname, [0.0, 0])
    a[0] += avg * n
    a[1] += n
# search_multi_kernel runs all strides of a step in one launch: bench.py's "search_kernel" kind times exactly that (the
# single-stride search_kernel launches of a run are its first step's, before any prefetch)
if "search_multi_kernel" in acc:
    acc["search_kernel"] = acc.pop("search_multi_kernel")
print(json.dumps({k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in sorted(acc.items())}, indent=1))
