"""developer: one forward of a wide layer on an instrumentation build (-DCONV3P_FW_ABLATE=32 prints per-wave stage ticks)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pointwise_amd import conv3p_op as op, synth
B, N, ci, co = 16, 4096, 36, 13
dev = torch.device("cuda:0")
P = synth.room_like(B, N, 40)
t = lambda a: torch.from_numpy(a).to(dev)
tp, tx, tw = t(P), t(synth.features(B, N, ci, 1, points=P)), t(synth.filter_weights(3, 3, 3, ci, co, 2))
cache = op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=27, max_cin=ci, max_cout=co)
op.conv3p(tp, tx, tw, (1, 1, 1), 0.1, cache=cache)
torch.cuda.synchronize()
