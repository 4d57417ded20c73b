"""developer (ON THE GPU BOX): per-phase times inside the workgroups of the FUSED forward launch (stack_forward_kernel) from an
instrumentation build (-DCONV3P_ABLATE=134217728: lane 0 of every wave of every 211th workgroup prints its 10-ns stamps; the
"loads" phase of layers 1.. contains the wait at the per-cloud barrier).
usage: CONV3P_HIP_LIB=devlibs/lib_stamps.so python tools/fused_trace.py"""
import os, sys, re, collections, subprocess
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    import torch
    from pointwise_amd import stack, synth
    dev = torch.device("cuda:0")
    B, N = 32, 2048
    P = torch.from_numpy(synth.modelnet_like(B, N, 40)).to(dev)
    st = stack.Conv3pStack(3, None, device=dev, seed=3, fused_launch={"none": False, "both": True}.get(sys.argv[2], sys.argv[2]))
    st.sparse_neighbourhoods = True
    ups = [torch.from_numpy(synth.upstream_grad(B, N, 9, 70 + i)).to(dev) for i in range(4)]
    for it in range(4):
        if it == 3:
            torch.cuda.synchronize(); print("==== last", flush=True)
        st.forward(P, P)
        st.backward(ups)
        torch.cuda.synchronize()
    sys.exit(0)
for mode in ("forward", "both", "none"):
    out = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", mode], capture_output=True, text=True).stdout
    out = out.split("==== last")[-1]
    for pat in ("fwd<3,9>", "fwd<9,9>", "bsp<9,9>", "fusedbwd wg"):
        rows = [l for l in out.splitlines() if l.startswith(pat)]
        if not rows:
            continue
        acc = collections.OrderedDict()
        for l in rows:
            for k, v in re.findall(r"([A-Za-z+\-]+\d?) +(\d+)(?= |$)", l.split(":", 1)[1]):
                acc.setdefault(k, []).append(int(v))
        print("%s, %s, %d waves; mean / max us: " % ({"forward": "forward fused", "both": "both passes fused", "none": "per-layer launches"}[mode], pat, len(rows)) +
              "  ".join("%s %.1f/%.1f" % (k, sum(v) / len(v) / 100, max(v) / 100) for k, v in acc.items()))
