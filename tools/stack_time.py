"""developer: time the models' conv3p stacks at other BASELINE shapes (cfg4: S3DIS scene_seg B=16 N=4096 C_in=9, 5 layers)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from pointwise_amd import stack, synth
dev = torch.device("cuda:0")
def run(name, B, N, cin, num_class, kind, steps=30):
    mk = synth.room_like if kind == "room" else synth.modelnet_like
    Ps = [torch.from_numpy(mk(B, N, 40 + i)).to(dev) for i in range(3)]
    Xs = [torch.from_numpy(synth.features(B, N, cin, 50 + i, points=p.cpu().numpy())).to(dev) for i, p in enumerate(Ps)]
    st = stack.Conv3pStack(cin, num_class, device=dev, seed=3)
    print('sparse neighbourhoods hint:', st.tune(Ps[0]))
    if os.environ.get('FORCE_HINT'): st.sparse_neighbourhoods = os.environ['FORCE_HINT'] == '1'; print('forced hint', st.sparse_neighbourhoods)
    nup = 1 if num_class is not None else 4
    cup = num_class if num_class is not None else stack.HIDDEN
    ups = [torch.from_numpy(synth.upstream_grad(B, N, cup, 60 + i)).to(dev) for i in range(nup)]
    def step(i):
        st.forward(Ps[i % 3], Xs[i % 3]); st.prefetch(Ps[(i + 1) % 3]); st.backward(ups)
    for i in range(5): step(i)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): step(i)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
    print("%s: B=%d N=%d C_in=%d layers=%d: %.3f ms/step fwd+bwd -> %.1f Mpoints/s" % (name, B, N, cin, len(st.layers), dt * 1e3, B * N / dt / 1e6))
if len(sys.argv) < 2: run("cfg2 classification", 32, 2048, 3, None, "modelnet")
run("cfg4 scene_seg", 16, 4096, 9, 13, "room")
if len(sys.argv) < 2: run("cfg4-like, N=8192", 8, 8192, 9, 13, "room")
