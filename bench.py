#!/usr/bin/env python
"""bench.py -- conv3p forward+backward throughput of the pointcnn2_acsd conv3p stack on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run,
one rank per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).

A "step" is one pass of the hot path over one batch of synthetic input: forward and backward of the four conv3p
layers of the classification model (3->9 s1, 9->9 s2, 9->9 s3, 9->9 s4, SELU in between;
/root/reference/pointcnn2_acsd.py:48-67) on ModelNet40-shaped clouds, B=32 clouds of N=2048 points PER GPU
(BASELINE.json configs[1]; weak scaling, configs[2] = 8 x 32), followed for N>1 by the one fused RCCL all-reduce of
the weight gradients.  Inputs are resident in HBM before the timed region.  metric = B*N*n_gpus / t_step.

Extra objects on the JSON line:
  roofline      dominant kernel (by HIP-event time, measured in a second, instrumented run of the same K steps
                on the same stream): algorithmic bytes per launch / average launch duration vs 8 TB/s HBM
  cpu_baseline  the oracle (CPU restatement of the reference op, OpenMP over the batch as the reference does)
                timed on this host's cores on a bounded sample of the same workload -- rank 0, N=1 only
  parity        max |delta| of y / dX / dW between the HIP path and the CPU oracle on a 2-cloud sample
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pointwise_amd import _lib, conv3p_op as op, distributed, stack, synth  # noqa: E402

B_PER_GPU = 32
N_POINTS = 2048
C_IN = 3
HBM_PEAK_GBS = 8000.0


def algorithmic_bytes(kind, B, N, cin, cout):
    """compulsory HBM bytes per launch (weights amortised), fp32."""
    pts = B * N
    if kind == "forward_kernel":
        return pts * (12 + 4 * cin + 4 * cout)
    if kind == "backward_kernel":
        return pts * (12 + 4 * cin + 4 * cout + 4 * cin)
    if kind == "search_kernel":
        # not one of the op's tensors-in/tensors-out kernels: its own compulsory traffic is the staged point
        # record in (16 B) and the per-tap populations out (4*27 B); pair lists are an implementation choice
        return pts * (16 + 4 * 27)
    return 0


def cpu_baseline(points_np, feats_np, st, ups_np, budget_s=12.0):
    """Time the CPU oracle on the same stack; also returns its outputs for the parity numbers."""
    from oracle import oracle                      # checker only; never on the product path
    threads = max(1, min(points_np.shape[0], os.cpu_count() or 1))
    filters = [f.detach().cpu().numpy() for f in st.filters]

    def one_pass(P, X, ups, nthreads):
        acts, x = [], X
        for li in range(4):
            s = st.layers[li][2]
            x = stack.selu_numpy(oracle.forward(P, x, filters[li], (s, s, s), stack.VOXEL, nthreads=nthreads))
            acts.append(x)
        carry, dws = None, [None] * 4
        for li in (3, 2, 1, 0):
            s = st.layers[li][2]
            g = ups[li] if carry is None else ups[li] + carry
            g = stack.selu_grad_numpy(acts[li], g)
            x_in = acts[li - 1] if li > 0 else X
            carry, dws[li] = oracle.backward(g, P, x_in, filters[li], (s, s, s), stack.VOXEL, nthreads=nthreads)
        return acts, carry, dws

    reps, t_total = 0, 0.0
    while t_total < budget_s and reps < 50:
        t0 = time.perf_counter()
        one_pass(points_np, feats_np, ups_np, threads)
        t_total += time.perf_counter() - t0
        reps += 1
    pts = points_np.shape[0] * points_np.shape[1]
    value = pts * reps / t_total / 1e6
    # serial (deterministic) oracle on 2 clouds for the parity numbers
    ref = one_pass(points_np[:2], feats_np[:2], [u[:2] for u in ups_np], 1)
    return {"value": round(value, 4), "unit": "Mpoints/s", "cores": threads, "kind": "port",
            "sample": "%d repetitions of the full workload (B=%d, N=%d, 4-layer stack fwd+bwd), OpenMP over "
                      "the batch like the reference, %.1f s of CPU time" % (reps, points_np.shape[0],
                                                                            points_np.shape[1], t_total)}, ref


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--serial", action="store_true",
                    help="developer: no side stream at all (clean per-kernel durations under rocprofv3)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="do not enqueue the next batch's neighbour search under the current batch's backward")
    args = ap.parse_args()

    rank, world, local = distributed.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    lib = _lib.load()

    # synthetic ModelNet40-shaped batches; features == points (modelnet_provider.py:212-213).
    # Every step gets a DIFFERENT batch (NBATCH distinct batches resident in HBM, cycled), as in training:
    # the neighbour cache therefore rebuilds its geometry every step (1 sort + 4 searches per step) and only
    # saves the repeats INSIDE a step (8 op calls share one `points`).  Nothing is carried across steps.
    NBATCH = 4
    Ps = [synth.modelnet_like(B_PER_GPU, N_POINTS, seed=1234 + 2 + 1000 * rank + 17 * i) for i in range(NBATCH)]
    P = Ps[0]
    ups_np = [synth.upstream_grad(B_PER_GPU, N_POINTS, stack.HIDDEN, 77 + li + 1000 * rank) for li in range(4)]
    tPs = [torch.from_numpy(p).to(dev) for p in Ps]
    tXs = [t.clone() for t in tPs]
    tP, tX = tPs[0], tXs[0]
    ups = [torch.from_numpy(u).to(dev) for u in ups_np]
    st = stack.Conv3pStack(C_IN, None, device=dev, seed=1234, overlap_search=not args.serial)
    counter = [0]
    # set-up, not a step: create the RCCL communicator (seconds on the first collective), allocate the stack's
    # neighbour caches and load the library's code object (one 64-point call) before anything is timed,
    # whatever --warmup is
    distributed.allreduce_weight_grads(torch.zeros_like(st.fused_grad))
    distributed.barrier()
    st.prepare(B_PER_GPU, N_POINTS)
    _p = tPs[0][:1, :64].contiguous()
    op.conv3p(_p, _p, st.filters[0], (1, 1, 1), stack.VOXEL)
    torch.cuda.synchronize(dev)

    prefetch = not (args.no_prefetch or args.serial)

    def step():
        i = counter[0] % NBATCH
        counter[0] += 1
        st.forward(tPs[i], tXs[i])
        if prefetch:
            # the next batch's geometry (1 sort + 4 searches) goes to the side stream now and runs under this
            # batch's backward, as a data-loader-fed training loop would do.  Every step still performs exactly
            # one full geometry build, one forward and one backward inside the timed region.
            st.prefetch(tPs[(i + 1) % NBATCH])
        dx, fused = st.backward(ups)
        distributed.allreduce_weight_grads(fused)
        return dx, fused

    # CPython's full (generation-2) collection walks every object torch and numpy created at import: a ~40 ms
    # host pause that hits once every few thousand allocations, i.e. somewhere inside a 100-step timed region
    # (tools/stall_probe.py).  Freeze the setup objects, as latency-sensitive Python loops do.
    import gc
    gc.collect()
    gc.freeze()
    for _ in range(args.warmup):
        step()
    distributed.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    t_enqueued = time.perf_counter() - t0       # host side done (diagnostic: is the step host- or GPU-bound?)
    torch.cuda.synchronize(dev)
    distributed.barrier()
    elapsed = distributed.max_over_ranks(time.perf_counter() - t0, dev)

    # ---- per-kernel HIP-event timing of the same K steps (instrumented, not the timed region) ----
    lib.conv3p_profile_reset()
    lib.conv3p_profile_enable(1)
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    lib.conv3p_profile_enable(0)
    import ctypes
    kinds = {}
    for k in range(lib.conv3p_profile_kinds()):
        n, ms = ctypes.c_uint64(0), ctypes.c_double(0.0)
        lib.conv3p_profile_read(k, ctypes.byref(n), ctypes.byref(ms))
        if n.value:
            kinds[lib.conv3p_profile_name(k).decode()] = (n.value, ms.value)
    lib.conv3p_profile_reset()

    out = None
    if rank == 0:
        total_pts = B_PER_GPU * N_POINTS * world
        ms_per_step = elapsed / args.steps * 1e3
        host_ms_per_step = t_enqueued / args.steps * 1e3
        dom = max(kinds.items(), key=lambda kv: kv[1][1])[0] if kinds else None
        roofline = None
        if dom is not None:
            n, ms = kinds[dom]
            # one real launch per layer and step does the work (the cache turns the repeats into early exits,
            # which stay in the duration sum): achieved = sum(bytes) / sum(time) over ALL launches of the kernel
            per_step_bytes = sum(algorithmic_bytes(dom, B_PER_GPU, N_POINTS, ci, co) for ci, co, _ in st.layers)
            launches_per_step = n / args.steps
            bytes_per_launch = per_step_bytes / max(launches_per_step, 1)
            avg_s = ms / n * 1e-3
            achieved = bytes_per_launch / avg_s / 1e9
            traffic = None
            tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
            if os.path.exists(tpath):
                try:
                    t = json.load(open(tpath)).get(dom)
                    # measured in separate rocprofv3 --pmc passes (tools/collect_profiles.sh): HBM bytes per launch,
                    # FETCH_SIZE raw + WRITE_SIZE (lower bound; see tools/traffic_json.py for the gfx950 caveat)
                    traffic = t["hbm_bytes_lower"] if isinstance(t, dict) else t
                except Exception:
                    traffic = None
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "avg_launch_us": round(avg_s * 1e6, 2), "bytes_per_launch": int(bytes_per_launch),
                        "kernel_ms_per_step": {k: round(v[1] / args.steps, 4) for k, v in kinds.items()}}
        out = {"metric": "conv3p fwd+bwd Mpoints/s", "value": round(total_pts / (elapsed / args.steps) / 1e6, 3),
               "unit": "Mpoints/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(ms_per_step, 4), "host_enqueue_ms_per_step": round(host_ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "cfg2 ModelNet40-shaped: B=32 clouds/GPU x N=2048, pointcnn2_acsd conv3p "
                                      "stack 3->9 s1, 9->9 s2, 9->9 s3, 9->9 s4 (+SELU), forward+backward, "
                                      "a different batch every step"
                                      + (", next batch's neighbour search enqueued on a side stream under the "
                                         "current backward" if prefetch else "")
                                      + (", fused RCCL all-reduce of 7290 weight grads" if world > 1 else ""),
                          "global_batch": B_PER_GPU * world, "points_per_cloud": N_POINTS,
                          "parallelism": "dp%d" % world},
               "roofline": roofline}
        if world == 1 and not args.no_cpu:
            base, ref = cpu_baseline(P, P.copy(), st, ups_np)
            out["cpu_baseline"] = base
            # parity of the HIP path against the oracle on the first two clouds
            acts = st.forward(tP[:2].contiguous(), tX[:2].contiguous())
            dx, fused = st.backward([u[:2].contiguous() for u in ups])
            torch.cuda.synchronize(dev)
            ref_acts, ref_dx, ref_dws = ref
            ref_fused = np.concatenate([w.reshape(-1) for w in ref_dws])
            out["parity"] = {
                "sample": "first 2 clouds of the workload, serial oracle",
                "max_abs_delta_y": float(max(np.abs(a.cpu().numpy() - r).max() for a, r in zip(acts, ref_acts))),
                "max_abs_delta_dX": float(np.abs(dx.cpu().numpy() - ref_dx).max()),
                "max_abs_delta_dW": float(np.abs(fused.cpu().numpy() - ref_fused).max()),
                "max_abs_dW": float(np.abs(ref_fused).max())}
        print(json.dumps(out), flush=True)
    distributed.barrier()


if __name__ == "__main__":
    main()
