"""developer: forward / backward time of one layer shape at a given cloud size (cached geometry), e.g.
   python tools/shape_time.py 36 13 16 4096 room [stride]   (a -DCONV3P_DEV_SKIP_SMALL build forces the matrix-core path)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from pointwise_amd import conv3p_op as op, synth
ci, co, B, N = [int(v) for v in sys.argv[1:5]]
kind = sys.argv[5] if len(sys.argv) > 5 else "room"
S = int(sys.argv[6]) if len(sys.argv) > 6 else 1
S3 = (S, S, S)
dev = torch.device("cuda:0")
P = (synth.room_like if kind == "room" else synth.modelnet_like)(B, N, 40)
t = lambda a: torch.from_numpy(a).to(dev)
tp, tx, tw, tdy = t(P), t(synth.features(B, N, ci, 1, points=P)), t(synth.filter_weights(3, 3, 3, ci, co, 2)), t(synth.upstream_grad(B, N, co, 3))
cache = op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=27, max_cin=ci, max_cout=co,
                         sparse_neighbourhoods=os.environ.get('FORCE_HINT') == '1')
f = lambda: op.conv3p(tp, tx, tw, S3, 0.1, cache=cache)
g = lambda: op.conv3p_grad(tdy, tp, tx, tw, S3, 0.1, cache=cache)
for fn in (f, g): fn(); fn()
def tm(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print("%d->%d B=%d N=%d %s stride %d lib=%s: forward %.3f ms, backward %.3f ms" % (ci, co, B, N, kind, S, os.path.basename(os.environ.get("CONV3P_HIP_LIB", "default")), tm(f), tm(g)))
