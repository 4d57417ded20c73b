"""SQ_INSTS_VALU (+ the other SQ counters of the same rocprofv3 --pmc pass) -> per-kernel averages per launch, keyed
like bench.py's HIP-event kinds (kernel base name without template arguments).
usage: python tools/valu_json.py <pmc_SQ_WAVES results.db> > profiles/valu_latest.json"""
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pointwise_amd.build import source_hash

db = sqlite3.connect(sys.argv[1])
cur = db.execute("select * from counters_collection limit 1")
cols = [d[0] for d in cur.description]
ki = "kernel_name" if "kernel_name" in cols else "name"
acc = {}
# the fused search is launched for ONE stencil by the bench's set-up / parity calls and for all FOUR by every step's
# prefetch: only the latter (the launches with the most waves) are "the step's search"
per_dispatch = {}
for did, name, cname, val in db.execute("select dispatch_id, %s, counter_name, value from counters_collection" % ki):
    if "search_fused_kernel" in name:
        per_dispatch.setdefault(did, {})[cname] = val
wmax = max((c.get("SQ_WAVES", 0) for c in per_dispatch.values()), default=0)
full = [c for c in per_dispatch.values() if c.get("SQ_WAVES", 0) == wmax]
for name, cname, n, avg in db.execute(
        "select %s, counter_name, count(*), avg(value) from counters_collection group by %s, counter_name" % (ki, ki)):
    if "conv3p::" not in name or "search_fused_kernel" in name:
        continue
    key = name.split("(")[0].replace("void ", "").replace("conv3p::", "").split("<")[0]
    key = {"prep_sort_kernel": "prep_kernel",
           "reduce_multi_kernel": "reduce_partials_kernel", "backward_sparse_kernel": "backward_kernel"}.get(key, key)
    a = acc.setdefault(key, {}).setdefault(cname, [0.0, 0])
    a[0] += avg * n
    a[1] += n
if full:
    for cname in full[0]:
        acc.setdefault("search_fused_kernel", {})[cname] = [sum(c.get(cname, 0.0) for c in full), len(full)]
# search_multi_kernel runs all strides of a step in one launch: bench.py's "search_kernel" kind times exactly that (the
# single-stride search_kernel launches of a run are its first step's, before any prefetch)
# the fused multi-stride search (round 4) likewise; tile_tables_kernel belongs to the same geometry step
for alias in ("search_fused_kernel", "search_multi_kernel"):
    if alias in acc:
        acc["search_kernel"] = acc.pop(alias)
        break
out = {k: {c: v[0] / v[1] for c, v in cs.items()} for k, cs in sorted(acc.items())}
out["_csrc_sha"] = source_hash()      # bench.py drops these counters when the library's sources have changed since
print(json.dumps(out, indent=1))
