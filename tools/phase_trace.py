"""developer (ON THE GPU BOX): per-phase times inside the workgroups of the 9 -> 9 forward and populated-rows backward
kernels from an instrumentation build (-DCONV3P_ABLATE=134217728 / -DCONV3P_SP_ABLATE=128: lane 0 of every wave of
every 211th workgroup prints its 10-ns stamps).  usage: CONV3P_HIP_LIB=devlibs/lib_x.so python tools/phase_trace.py [stride]"""
import os, sys, re, collections, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    from pointwise_amd import conv3p_op as op, synth
    S = int(sys.argv[2]); CI = int(sys.argv[3]) if len(sys.argv) > 3 else 9
    dev = torch.device("cuda:0")
    B, N, ci, co = 32, 2048, CI, 9
    P = synth.modelnet_like(B, N, 40)
    t = lambda a: torch.from_numpy(a).to(dev)
    tp, tx, tw, tdy = t(P), t(synth.features(B, N, ci, 1, points=P)), t(synth.filter_weights(3, 3, 3, ci, co, 2)), t(synth.upstream_grad(B, N, co, 3))
    cache = op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=27, max_cin=ci, max_cout=co)
    for it in range(4):
        if it == 3:
            torch.cuda.synchronize(); print("==== last", flush=True)
        op.conv3p(tp, tx, tw, (S, S, S), 0.1, cache=cache)
        op.conv3p_grad(tdy, tp, tx, tw, (S, S, S), 0.1, cache=cache)
        torch.cuda.synchronize()
    sys.exit(0)
S = sys.argv[1] if len(sys.argv) > 1 else "2"
CIARG = sys.argv[2] if len(sys.argv) > 2 else "9"
out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", S, CIARG], capture_output=True, text=True).stdout
out = out.split("==== last")[-1]
for pat in ("fwd<", "bsp<", "bwd<"):
    rows = [l for l in out.splitlines() if l.startswith(pat)]
    if not rows:
        continue
    acc = collections.OrderedDict()
    for l in rows:
        for k, v in re.findall(r"([A-Za-z+\-]+) (\d+)(?= |$)", l.split(":", 1)[1]):
            acc.setdefault(k, []).append(int(v))
    print("%s stride %s, %d waves; mean / max us: " % (pat, S, len(rows)) + "  ".join("%s %.1f/%.1f" % (k, sum(v) / len(v) / 100, max(v) / 100) for k, v in acc.items()))
