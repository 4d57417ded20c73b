"""Data parallelism for the conv3p hot path: shard clouds across GPUs, all-reduce the weight gradients.

Every cloud is independent in the forward pass and in grad_input (the reference itself parallelises over the
batch, /root/reference/tf_ops/conv3p/tf_conv3p_atrous.cpp:453-456, :620-622); the only cross-cloud quantity is
grad_filter (.cpp:696, :709-716).  So: one process per GPU, contiguous batch shards, replicated filters, and ONE
sum all-reduce (RCCL over xGMI; torch.distributed backend "nccl" on ROCm) over a single fused buffer holding all
layers' grad_filter.  Sum, not mean: any 1/B loss scaling already lives in the upstream gradient.
The fused buffer is 29 KB for the ModelNet40 stack (latency-bound on xGMI: one call, never one per layer) and
3.5 MB for a 128->256 layer.
"""
import os

import torch
import torch.distributed as dist


def shard_bounds(batch, world_size, rank):
    """Contiguous [lo, hi) shard of `batch` clouds for `rank`; remainders go to the first ranks."""
    if world_size <= 0 or not (0 <= rank < world_size):
        raise ValueError("bad world_size/rank")
    base, rem = divmod(batch, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT (torchrun's contract).
    Returns (rank, world_size, local_rank).  Single-process runs do not create a process group."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def allreduce_weight_grads(fused_grad, group=None):
    """In-place sum all-reduce of the fused grad_filter buffer (no-op without a process group)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(fused_grad, op=dist.ReduceOp.SUM, group=group)
    return fused_grad


def shard_range(numel, world_size, rank):
    """[lo, hi) of the flat parameter slice `rank` owns: equal slices of ceil(numel / world) elements (the last
    ones may be shorter or empty), the layout reduce_scatter / all_gather use."""
    per = (numel + world_size - 1) // world_size
    lo = min(numel, rank * per)
    return lo, min(numel, lo + per)


def reduce_scatter_grad(grad, group=None):
    """Sum-reduce a large flat gradient (the head's 151 MB dW1) so that every rank ends up with the SUM of its own
    1/world slice only: half the traffic of an all-reduce.  Returns (slice tensor, lo, hi).  RCCL: one
    reduce_scatter over xGMI; backends without it (gloo, used by the CPU tests): all-reduce, then slice."""
    flat = grad.reshape(-1)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return flat, 0, flat.numel()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = (flat.numel() + world - 1) // world
    lo, hi = shard_range(flat.numel(), world, rank)
    if dist.get_backend(group) == "nccl":
        pad = per * world - flat.numel()
        src = flat if pad == 0 else torch.cat([flat, flat.new_zeros(pad)])
        out = torch.empty(per, dtype=flat.dtype, device=flat.device)
        dist.reduce_scatter_tensor(out, src, op=dist.ReduceOp.SUM, group=group)
        return out[:hi - lo], lo, hi
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    return flat[lo:hi], lo, hi


def all_gather_param(param, shard, group=None):
    """Inverse of the sharding above: every rank contributes its updated slice, all ranks end with the full `param`
    (updated in place)."""
    flat = param.reshape(-1)
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        if shard.data_ptr() != flat.data_ptr():
            flat.copy_(shard)
        return param
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = (flat.numel() + world - 1) // world
    mine = flat.new_zeros(per)
    mine[:shard.numel()] = shard
    if dist.get_backend(group) == "nccl":
        full = torch.empty(per * world, dtype=flat.dtype, device=flat.device)
        dist.all_gather_into_tensor(full, mine, group=group)
    else:
        parts = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(parts, mine, group=group)
        full = torch.cat(parts)
    flat.copy_(full[:flat.numel()])
    return param


def sharded_momentum_step(param, grad, momentum_shard, lr, momentum=0.9, group=None):
    """One data-parallel step for a LARGE parameter (the head's fc1 matrix): reduce-scatter the gradient, apply
    tf.train.MomentumOptimizer's rule (accum = momentum * accum + grad; param -= lr * accum;
    /root/reference/train_modelnet40_acsd.py:81) to this rank's slice only -- momentum_shard holds just that slice, so
    optimizer state is 1/world per GPU -- and all-gather the updated weights."""
    g, lo, hi = reduce_scatter_grad(grad, group)
    momentum_shard.mul_(momentum).add_(g)
    new = param.reshape(-1)[lo:hi] - lr * momentum_shard
    return all_gather_param(param, new, group)


def max_over_ranks(value, device):
    """MAX of a python float over ranks (bench timing contract)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        t = torch.tensor([value], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())
    return value


def barrier():
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
