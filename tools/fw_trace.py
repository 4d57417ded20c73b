"""developer: one forward of a layer on an instrumentation build (-DCONV3P_ABLATE=134217728 prints per-wave stage times)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pointwise_amd import conv3p_op as op, synth
ci, co, B, N = [int(v) for v in sys.argv[1:5]]
kind = sys.argv[5] if len(sys.argv) > 5 else "room"
S = int(sys.argv[6]) if len(sys.argv) > 6 else 1
dev = torch.device("cuda:0")
P = (synth.room_like if kind == "room" else synth.modelnet_like)(B, N, 40)
t = lambda a: torch.from_numpy(a).to(dev)
tp, tx, tw = t(P), t(synth.features(B, N, ci, 1, points=P)), t(synth.filter_weights(3, 3, 3, ci, co, 2))
cache = op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=27, max_cin=ci, max_cout=co)
op.cache_prepare(tp, (3, 3, 3), (S, S, S), 0.1, cache)
torch.cuda.synchronize()
print("---- forward")
op.conv3p(tp, tx, tw, (S, S, S), 0.1, cache=cache, points_unchanged=True)
torch.cuda.synchronize()
