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


def _stack_worker(rank, world, port, B, out_dir):
    """Each rank fills a real Conv3pStack's fused gradient buffer through its grad_views (the layout the stack-level
    backward writes into) from its shard, then the one fused all-reduce; plus a cfg5-sized (884 736 floats) buffer."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from oracle import oracle
    from pointwise_amd import distributed, stack, synth
    distributed.init_from_env(backend="gloo")
    st = stack.Conv3pStack(3, 13, device="cpu", dtype=torch.float64, seed=9)      # 5 layers incl. the 36 -> 13 head
    assert st.fused_grad.numel() == sum(27 * ci * co for ci, co, _ in st.layers)
    lo, hi = distributed.shard_bounds(B, world, rank)
    P = synth.modelnet_like(B, 96, seed=70).astype(np.float64)
    for li, (ci, co, s) in enumerate(st.layers):
        X = synth.features(B, 96, ci, 71 + li, dtype=np.float64)
        dY = synth.upstream_grad(B, 96, co, 81 + li, dtype=np.float64)
        _, dw = oracle.backward(dY[lo:hi], P[lo:hi], X[lo:hi], st.filters[li].numpy(), (s, s, s), 0.1)
        st.grad_views[li].copy_(torch.from_numpy(dw))          # what the backward kernels do on the device
    distributed.allreduce_weight_grads(st.fused_grad)
    big = torch.full((27 * 128 * 256,), float(rank + 1), dtype=torch.float32)   # cfg5's grad_filter: 3.54 MB
    big[rank::7] += 0.5
    distributed.allreduce_weight_grads(big)
    np.save(os.path.join(out_dir, "stack_%d.npy" % rank), st.fused_grad.numpy())
    np.save(os.path.join(out_dir, "big_%d.npy" % rank), big.numpy())
    dist.destroy_process_group()


def test_stack_fused_buffer_layout_and_cfg5_sized_allreduce(tmp_path):
    world, B = 2, 5
    port = _free_port()
    mp.spawn(_stack_worker, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from oracle import oracle
    from pointwise_amd import stack, synth
    st = stack.Conv3pStack(3, 13, device="cpu", dtype=torch.float64, seed=9)
    P = synth.modelnet_like(B, 96, seed=70).astype(np.float64)
    full = []
    for li, (ci, co, s) in enumerate(st.layers):
        X = synth.features(B, 96, ci, 71 + li, dtype=np.float64)
        dY = synth.upstream_grad(B, 96, co, 81 + li, dtype=np.float64)
        full.append(oracle.backward(dY, P, X, st.filters[li].numpy(), (s, s, s), 0.1)[1].reshape(-1))
    full = np.concatenate(full)
    want_big = np.full(27 * 128 * 256, 3.0, dtype=np.float32)
    want_big[0::7] += 0.5
    want_big[1::7] += 0.5
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "stack_%d.npy" % r))
        assert np.abs(got - full).max() <= 1e-12 * max(1.0, np.abs(full).max())
        assert np.array_equal(np.load(os.path.join(str(tmp_path), "big_%d.npy" % r)), want_big)


def _shard_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from pointwise_amd import distributed
    distributed.init_from_env(backend="gloo")
    g = torch.Generator().manual_seed(11)
    W = torch.randn(1001, 8, generator=g, dtype=torch.float64)          # numel not divisible by the world size
    grads = [torch.randn(1001, 8, generator=g, dtype=torch.float64) for _ in range(world)]
    lo, hi = distributed.shard_range(W.numel(), world, rank)
    mom = torch.zeros(hi - lo, dtype=torch.float64)
    for _ in range(2):                                                   # two steps: the momentum state carries over
        distributed.sharded_momentum_step(W, grads[rank].clone(), mom, lr=0.1, momentum=0.9)
    np.save(os.path.join(out_dir, "w_%d.npy" % rank), W.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_momentum_step_equals_all_reduce_training(tmp_path, world):
    """Reduce-scatter + sharded momentum update + all-gather of a large parameter reproduces the plain
    all-reduce + full update on every rank (the head's fc1 path in data-parallel training)."""
    port = _free_port()
    mp.spawn(_shard_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    g = torch.Generator().manual_seed(11)
    W = torch.randn(1001, 8, generator=g, dtype=torch.float64)
    grads = [torch.randn(1001, 8, generator=g, dtype=torch.float64) for _ in range(world)]
    total = sum(grads)
    acc = torch.zeros_like(W)
    for _ in range(2):
        acc = 0.9 * acc + total
        W = W - 0.1 * acc
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "w_%d.npy" % r))
        assert np.abs(got - W.numpy()).max() <= 1e-12


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


def _bench_step_worker(rank, world, port, B, out_dir):
    """bench.py's step assembly at world 2 on CPU tensors: shard the batch -> the shard's backward writes every
    layer's grad_filter into the stack's grad_views -> bench.Reducer launches the ONE fused all-reduce (timed) ->
    the JSON fields the N > 1 line carries.  Also the head's loss scaling under SUM-reduced gradients."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    import json
    import bench
    from oracle import oracle
    from pointwise_amd import distributed, head, stack, synth
    distributed.init_from_env(backend="gloo")
    st = stack.Conv3pStack(3, None, device="cpu", dtype=torch.float64, seed=21)
    lo, hi = distributed.shard_bounds(B, world, rank)
    P = synth.modelnet_like(B, 80, seed=90).astype(np.float64)
    red = bench.Reducer("cpu", world)
    red.timing = True
    steps = 2
    for _ in range(steps):
        red.wait_previous()
        for li, (ci, co, s) in enumerate(st.layers):
            X = synth.features(B, 80, ci, 91 + li, dtype=np.float64)
            dY = synth.upstream_grad(B, 80, co, 95 + li, dtype=np.float64)
            _, dw = oracle.backward(dY[lo:hi], P[lo:hi], X[lo:hi], st.filters[li].numpy(), (s, s, s), 0.1)
            st.grad_views[li].copy_(torch.from_numpy(dw))
        red.launch(st.fused_grad)
    red.finish()
    spread = bench.rank_spread(0.010 * (rank + 1), steps, torch.device("cpu"))     # rank r "took" 10 (r + 1) ms
    fields = {"rccl_world": dist.get_world_size(), "allreduce_ms_per_step": red.ms_per_step(steps),
              "allreduce_exposed_ms_per_step": red.exposed_ms_per_step(steps), "ms_per_step_ranks": spread,
              "allreduce_bytes": int(st.fused_grad.numel() * 8)}
    json.dumps(fields)
    assert fields["allreduce_exposed_ms_per_step"] == fields["allreduce_ms_per_step"]   # synchronous on CPU tensors
    assert abs(spread["min"] - 10.0 / steps) < 1e-6 and abs(spread["max"] - 10.0 * world / steps) < 1e-6
    np.save(os.path.join(out_dir, "bench_fused_%d.npy" % rank), st.fused_grad.numpy())
    np.save(os.path.join(out_dir, "bench_fields_%d.npy" % rank),
            np.asarray([fields["rccl_world"], fields["allreduce_ms_per_step"], fields["allreduce_bytes"]], dtype=np.float64))
    # head: each rank's dlogits / GLOBAL batch, gradients summed over ranks == the full batch's mean-loss gradient
    g = torch.Generator().manual_seed(5)
    logits = torch.randn((B, 7), generator=g, dtype=torch.float64)
    labels = torch.randint(0, 7, (B,), generator=g)
    hd = head.ClassificationHead.__new__(head.ClassificationHead)      # loss() uses no state
    _, dl = hd.loss(logits[lo:hi], labels[lo:hi], global_batch=B)
    full = torch.zeros((B, 7), dtype=torch.float64)
    full[lo:hi] = dl
    dist.all_reduce(full)
    np.save(os.path.join(out_dir, "dlogits_%d.npy" % rank), full.numpy())
    dist.destroy_process_group()


def test_bench_step_assembly_and_loss_scaling_at_world_2(tmp_path):
    world, B = 2, 5
    port = _free_port()
    mp.spawn(_bench_step_worker, args=(world, port, B, str(tmp_path)), nprocs=world, join=True)
    sys.path.insert(0, ROOT)
    from oracle import oracle
    from pointwise_amd import head, stack, synth
    st = stack.Conv3pStack(3, None, device="cpu", dtype=torch.float64, seed=21)
    P = synth.modelnet_like(B, 80, seed=90).astype(np.float64)
    full = []
    for li, (ci, co, s) in enumerate(st.layers):
        X = synth.features(B, 80, ci, 91 + li, dtype=np.float64)
        dY = synth.upstream_grad(B, 80, co, 95 + li, dtype=np.float64)
        full.append(oracle.backward(dY, P, X, st.filters[li].numpy(), (s, s, s), 0.1)[1].reshape(-1))
    full = np.concatenate(full)
    g = torch.Generator().manual_seed(5)
    logits = torch.randn((B, 7), generator=g, dtype=torch.float64)
    labels = torch.randint(0, 7, (B,), generator=g)
    hd = head.ClassificationHead.__new__(head.ClassificationHead)
    _, want_dl = hd.loss(logits, labels)                                # single process, full batch
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), "bench_fused_%d.npy" % r))
        assert np.abs(got - full).max() <= 1e-12 * max(1.0, np.abs(full).max())
        f = np.load(os.path.join(str(tmp_path), "bench_fields_%d.npy" % r))
        assert f[0] == world and f[1] > 0.0 and f[2] == full.size * 8
        assert np.abs(np.load(os.path.join(str(tmp_path), "dlogits_%d.npy" % r)) - want_dl.numpy()).max() <= 1e-14
