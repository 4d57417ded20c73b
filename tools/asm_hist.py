"""developer: instruction histogram of one kernel in a device assembly listing
(hipcc --offload-arch=gfx950 ... --cuda-device-only -S -o lib.s pointwise_amd/csrc/conv3p_abi.hip).
usage: python tools/asm_hist.py lib.s <mangled-name-substring> [--loop]   (--loop: the largest loop body only)"""
import collections, re, sys
s = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2]
i0 = next(i for i, l in enumerate(s) if l.startswith("_Z") and pat in l and l.split(":")[0].endswith(l.split(":")[0]) and ":" in l and not l.startswith("\t"))
i1 = next(i for i in range(i0, len(s)) if s[i].startswith(".Lfunc_end"))
body = s[i0:i1]
def hist(lines):
    c = collections.Counter()
    for l in lines:
        l = l.strip()
        if not l or l.startswith((".", ";", "_")) or l.endswith(":"):
            continue
        c[l.split()[0]] += 1
    return c
c = hist(body)
print(len(body), "lines;", sum(c.values()), "instructions")
grp = collections.Counter()
for k, v in c.items():
    g = "mfma" if "mfma" in k else k.split("_")[0] + ("_" + k.split("_")[1] if k.startswith(("ds_", "global_", "buffer_", "scratch_")) else "")
    grp[g] += v
print(sorted(grp.items(), key=lambda x: -x[1]))
print(c.most_common(40))
for l in s[i1:i1 + 80]:
    if any(t in l for t in ("vgpr_count", "vgpr_spill", "sgpr_count", "lds_size", "NumVgprs", "ScratchSize", "Occupancy", "NumAgprs")):
        print(l.strip())
