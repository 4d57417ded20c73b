"""SQ_INSTS_MFMA per launch of the matrix-core path's kernels (tools/pmc_query.py summary of the deep PMC pass) ->
profiles/deep_mfma_latest.json, stamped with the sha of the library's sources; bench.py turns it into the cfg5
roofline's `frac_issued`.   usage: python tools/mfma_json.py <r05_deep_pmc_MFMA.txt> > profiles/deep_mfma_latest.json"""
import json, os, re, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pointwise_amd.build import source_hash
tot, per = 0.0, {}
for l in open(sys.argv[1]):
    m = re.match(r"(.*?)\s+SQ_INSTS_MFMA\s+n=(\d+)\s+avg=([0-9.e+]+)", l)
    if m and float(m.group(3)) > 0:
        per[re.search(r"(deep_\w+<[^>]*>?)", m.group(1)).group(1)] = float(m.group(3))
        tot += float(m.group(3))
print(json.dumps({"workload": "cfg5 shard step (tools/deep_time.py): B=16 x N=8192, 128->256, forward + backward",
                  "mfma_instructions_per_step": tot, "per_kernel": per, "flops_per_instruction": 4096,
                  "note": "SQ_INSTS_MFMA per launch (rocprofv3 --pmc); v_mfma_f32_32x32x2_f32 = 2*32*32*2 flops",
                  "_csrc_sha": source_hash()}, indent=1))
