"""developer (ON THE GPU BOX): the cfg4 step (S3DIS scene_seg stack, B=16 x N=4096, prefetch on the side stream) alone;
prints ms/step.  Under rocprofv3 --kernel-trace: tools/timeline.py."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import bench
from pointwise_amd import synth, stack

dev = torch.device("cuda:0")
B, N, cin, ncls = 16, 4096, 9, 13
Ps = [torch.from_numpy(synth.room_like(B, N, 40 + i)).to(dev) for i in range(3)]
Xs = [torch.from_numpy(synth.features(B, N, cin, 50 + i, points=p.cpu().numpy())).to(dev) for i, p in enumerate(Ps)]
st = stack.Conv3pStack(cin, ncls, device=dev, seed=3, fused_launch="--fused" in sys.argv)
if "--no-tune" not in sys.argv:
    st.tune(Ps[0])
if "--sparse" in sys.argv:   # developer: the populated-rows backward for the dilated layers whatever the lists look like
    st.sparse_neighbourhoods = True
    for c in st._caches:
        if c is not None:
            c.sparse_neighbourhoods = True
ups = [torch.from_numpy(synth.upstream_grad(B, N, ncls, 60)).to(dev)]
ctr = [0]
pre = "--no-prefetch" not in sys.argv
first = "--prefetch-first" in sys.argv
def step():
    i = ctr[0] % 3
    ctr[0] += 1
    if pre and first:
        st.prefetch(Ps[(i + 1) % 3])      # developer: the next batch's geometry under this batch's FORWARD
    st.forward(Ps[i], Xs[i])
    if pre and not first:
        st.prefetch(Ps[(i + 1) % 3])
    st.backward(ups)
print("cfg4 step: %.4f ms" % (bench.timed(dev, step, 20, 5) * 1e3))
