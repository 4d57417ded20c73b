"""developer (ON THE GPU BOX, with a -DCONV3P_DEV_WALK_STATS build: CONV3P_HIP_LIB=devlibs/lib_walkstats.so): lane use of
the list-walking kernels on the cfg2 and cfg4 workloads -- one forward pass of each stack; the library prints one
`walk_stats` line per forward launch to stderr (see dev_walk_stats in conv3p_abi.hip)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pointwise_amd import stack, synth
dev = torch.device("cuda:0")
for name, B, N, cin, ncls, mk in (("cfg2", 32, 2048, 3, None, synth.modelnet_like), ("cfg4", 16, 4096, 9, 13, synth.room_like)):
    P = torch.from_numpy(mk(B, N, 40)).to(dev)
    X = torch.from_numpy(synth.features(B, N, cin, 50, points=P.cpu().numpy())).to(dev)
    st = stack.Conv3pStack(cin, ncls, device=dev, seed=3)
    print("==", name, file=sys.stderr, flush=True)
    st.forward(P, X)
    torch.cuda.synchronize()
