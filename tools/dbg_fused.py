"""developer (ON THE GPU BOX): fused stack launches step by step with a synchronisation after each call."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from pointwise_amd import stack, synth
dev = torch.device("cuda:0")
def run(cin, ncls, B, N, kind, tune):
    P = synth.modelnet_like(B, N, seed=5) if kind == "modelnet" else synth.room_like(B, N, 5)
    X = synth.features(B, N, cin, 6, points=P)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    tp, tx = t(P), t(X)
    res = []
    for c_stack in (True, False):
        st = stack.Conv3pStack(cin, ncls, device=dev, seed=3, c_stack=c_stack, fused_launch=c_stack)
        if tune:
            st.tune(tp)
        print("  c_stack", c_stack, "sparse", st.sparse_neighbourhoods, flush=True)
        acts = st.forward(tp, tx)
        torch.cuda.synchronize(); print("  forward ok", flush=True)
        nl = 1 if ncls else 4
        ups = [t(synth.upstream_grad(B, N, ncls if ncls else 9, 70 + i)) for i in range(nl)]
        dx, fused = st.backward(ups)
        torch.cuda.synchronize(); print("  backward ok", flush=True)
        print("  fused status", st.fused_status(), flush=True)
        res.append(([a.clone() for a in acts], dx.clone(), fused.clone()))
    for a, b in zip(res[0][0], res[1][0]):
        print("  act equal", torch.equal(a, b), float((a - b).abs().max()))
    print("  dx equal", torch.equal(res[0][1], res[1][1]), float((res[0][1] - res[1][1]).abs().max()),
          " dW max rel", float((res[0][2] - res[1][2]).abs().max() / res[1][2].abs().max()))
for args in ((3, None, 4, 512, "modelnet", True), (3, None, 32, 2048, "modelnet", True), (9, 13, 2, 4096, "room", False), (9, 13, 16, 4096, "room", False),
             (9, 13, 2, 4096, "room", True), (3, None, 3, 700, "modelnet", True)):
    print(args, flush=True)
    run(*args)
