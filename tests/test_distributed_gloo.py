"""world_size-2 test of the data-parallel path on CPU (gloo): contiguous batch shards + ONE fused sum
all-reduce of the weight gradients reproduce the full-batch grad_filter.  The per-shard compute is done by the
CPU oracle here (tests may use it; there is no GPU in this container) -- what is under test is the host logic in
pointwise_amd.distributed: shard bounds, the fused buffer, the collective, the max-over-ranks timing helper."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from oracle import oracle
    from pointwise_amd import distributed
    from tests.parity_util import make_case
    r, w, _ = distributed.init_from_env(backend="gloo")
    assert (r, w) == (rank, world)
    layers = [(3, 9, 1), (9, 9, 2)]
    lo, hi = distributed.shard_bounds(B, world, rank)
    sizes = [27 * ci * co for ci, co, _ in layers]
    fused = torch.zeros(sum(sizes), dtype=torch.float64)
    o = 0
    for li, (ci, co, s) in enumerate(layers):
        P, X, W, dY = make_case("modelnet", B, 128, ci, co, seed=60 + li, dtype=np.float64)
        _, dw = oracle.backward(dY[lo:hi], P[lo:hi], X[lo:hi], W, (s, s, s), 0.1)
        fused[o:o + sizes[li]] = torch.from_numpy(dw.reshape(-1))
        o += sizes[li]
    distributed.allreduce_weight_grads(fused)
    t = distributed.max_over_ranks(float(rank + 1), torch.device("cpu"))
    distributed.barrier()
    np.save(os.path.join(out_dir, "fused_%d.npy" % rank), fused.numpy())
    np.save(os.path.join(out_dir, "tmax_%d.npy" % rank), np.asarray(t))
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [6, 5])
def test_sharded_weight_grads_allreduce_to_full_batch(tmp_path, B):
    world = 2
    port = _free_port()
    mp.spawn(_worker, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from oracle import oracle
    from tests.parity_util import make_case
    full = []
    for li, (ci, co, s) in enumerate([(3, 9, 1), (9, 9, 2)]):
        P, X, W, dY = make_case("modelnet", B, 128, ci, co, seed=60 + li, dtype=np.float64)
        full.append(oracle.backward(dY, P, X, W, (s, s, s), 0.1)[1].reshape(-1))
    full = np.concatenate(full)
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "fused_%d.npy" % r))
        assert np.abs(got - full).max() <= 1e-12 * max(1.0, np.abs(full).max())
        assert float(np.load(os.path.join(str(tmp_path), "tmax_%d.npy" % r))) == 2.0
