#!/usr/bin/env python
"""bench.py -- conv3p forward+backward throughput of the pointcnn2_acsd conv3p stack on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 with WORLD_SIZE unset (plain `python bench.py --gpus 8`): bench.py re-executes itself through
  `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`, one rank per GPU;
  launched by torch.distributed.run it reads RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.
  Fewer than N visible devices, or WORLD_SIZE != N, is an error (never a silent n_gpus = 1 run).

A "step" is one pass of the hot path over one batch of synthetic input: forward and backward of the four conv3p
layers of the classification model (3->9 s1, 9->9 s2, 9->9 s3, 9->9 s4, SELU in between;
/root/reference/pointcnn2_acsd.py:48-67) on ModelNet40-shaped clouds, B=32 clouds of N=2048 points PER GPU
(BASELINE.json configs[1]; weak scaling, configs[2] = 8 x 32), followed for N>1 by the one fused RCCL all-reduce of
the weight gradients.  Inputs are resident in HBM before the timed region.  metric = B*N*n_gpus / t_step.

Extra objects on the JSON line (rank 0; everything except `roofline` only at N=1):
  roofline           dominant kernel of the headline step (by HIP-event time, measured live in a second,
                     instrumented run of the same K steps on the same streams): algorithmic bytes per launch /
                     average launch duration vs 8 TB/s.  avg_launch_us is measured WITH the side-stream overlap the
                     headline uses (other kernels share the CUs); avg_launch_us_isolated comes from a third pass
                     with no side stream at all -- `achieved` / `frac` use the overlapped figure (the conservative one).
  cpu_baseline       the REFERENCE's own Conv3p / Conv3pGrad batch loops (oracle/_ref/libref_compute_atrous_omp.so:
                     tf_conv3p_atrous.cpp compiled in place with the reference's flags -O3 -fopenmp -DCONV_OPENMP)
                     timed on this host's cores on a bounded sample of the same workload; kind = "reference".
                     Falls back to the oracle's C restatement (kind = "port") where oracle/_ref is absent.
  cpu_baseline_cfg1  BASELINE config 1: B=1, N=2048, 3->9, stride 1, forward only, serial reference loops.
  stateless_ms_per_step   the same cfg2 step through the stateless C entry points exactly as the TF shim
                     (integration/tf_conv3p_shim.cc) calls them: conv3p_forward_f32 / conv3p_backward_f32 per op, no
                     cache, no hints, no prefetch, SELU as its own op -- what an unmodified TF graph would get.
  op_boundary_cached_ms_per_step   the same 8 op calls through conv3p_forward/backward_cached_* with a persistent
                     neighbour cache but NO caller hints and no prefetch (every call re-validates the points on the
                     device): what a TF shim holding persistent state gets without owning the step loop.
  op_boundary_native_ms_per_step   that op-by-op step once more WITHOUT Python: integration/op_boundary_bench (C++, the
                     library's C ABI only) issues the 8 *_cached_* calls and the SELU ops on one stream over the same
                     clouds, HIP-event timed; run as a subprocess outside every timed region.
  valu_issue_frac    (inside roofline) per kernel: SQ_INSTS_VALU per launch (rocprofv3 PMC pass, profiles/valu_latest.json)
                     / (1024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction x launch duration) -- the resource the
                     cache-resident cfg2 kernels actually load, HBM being nowhere near saturated.
  --workload cfg5    B=16 x N=8192 clouds PER GPU, one 128->256 layer, forward+backward, one 3.54 MB all-reduce.
  N > 1              the all-reduce runs on its own communication stream (the next step's prefetch / forward start
                     under it), is timed with HIP events on that stream (allreduce_ms_per_step), and rccl_world is
                     what torch.distributed reports after the first collective.
  other_configs      cfg4 (S3DIS scene_seg stack, B=16 x N=4096, 5 layers) and the cfg5 per-GPU shard (B=16 x N=8192,
                     one 128->256 layer): ms/step, Mpoints/s and a roofline each (cfg4 HBM, 1376 B/point;
                     cfg5 fp32 MFMA, USEFUL flops = 5.31 MFLOP/point = 3 x 2*27*Cin*Cout, vs 157.3 TFLOP/s).
  parity             max |delta| of y / dX / dW between the HIP path and the CPU oracle on a 2-cloud sample
"""
import argparse
import ctypes
import json
import os
import socket
import subprocess
import sys
import time

os.environ.setdefault("OMP_WAIT_POLICY", "passive")   # CPU legs: no spinning between the short parallel regions

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pointwise_amd import _lib, conv3p_op as op, distributed, stack, synth  # noqa: E402

B_PER_GPU = 32
N_POINTS = 2048
C_IN = 3
HBM_PEAK_GBS = 8000.0
MFMA_F32_PEAK_TFLOPS = 157.3


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


def stack_bytes_per_point(layers):
    """fwd+bwd = 24 + 12*Cin + 8*Cout bytes per point and layer."""
    return sum(24 + 12 * ci + 8 * co for ci, co, _ in layers)


def read_kernel_times(lib):
    kinds = {}
    for k in range(lib.conv3p_profile_kinds()):
        n, ms = ctypes.c_uint64(0), ctypes.c_double(0.0)
        lib.conv3p_profile_read(k, ctypes.byref(n), ctypes.byref(ms))
        if n.value:
            kinds[lib.conv3p_profile_name(k).decode()] = (n.value, ms.value)
    lib.conv3p_profile_reset()
    return kinds


def profile_steps(lib, dev, step, steps):
    """Per-kernel HIP-event timing of `steps` steps (instrumented; never the timed region)."""
    lib.conv3p_profile_reset()
    lib.conv3p_profile_enable(1)
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    lib.conv3p_profile_enable(0)
    return read_kernel_times(lib)


def timed(dev, step, steps, warmup):
    for _ in range(warmup):
        step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps


SIMDS = 1024
CLOCK_GHZ = 2.4


def load_counters(name):
    """profiles/<name>_latest.json (collected in separate rocprofv3 --pmc passes, tools/collect_profiles.sh) if it was
    collected for the kernels this run loads: the files carry the sha of the library's sources (`_csrc_sha`,
    pointwise_amd.build.source_hash()).  Returns (table or None, stale?)."""
    path = os.path.join(ROOT, "profiles", name + "_latest.json")
    if not os.path.exists(path):
        return None, False
    try:
        table = json.load(open(path))
    except Exception:
        return None, False
    from pointwise_amd.build import source_hash
    try:
        now = source_hash()
    except Exception:
        now = None
    if table.get("_csrc_sha") != now:
        return None, True
    return table, False


def valu_issue_fracs(kinds, steps):
    """Per kernel: fraction of the chip's vector-issue slots its launches use (see the module docstring)."""
    table, _ = load_counters("valu")
    if table is None:
        return None
    out = {}
    for k, (n, ms) in kinds.items():
        rec = table.get(k)
        if not rec or not n or ms <= 0:
            continue
        insts = rec["SQ_INSTS_VALU"]                       # wave instructions per launch, averaged over its launches
        slots = SIMDS * CLOCK_GHZ * 1e9 / 4.0 * (ms / n * 1e-3)
        out[k] = {"insts_per_launch": int(insts), "avg_launch_us": round(ms / n * 1e3, 2), "frac": round(insts / slots, 4)}
    return out


class Reducer:
    """The fused weight-gradient all-reduce of a step, on its own stream, timed with HIP events on that stream."""

    def __init__(self, dev, world):
        self.world = world
        self.dev = torch.device(dev)
        self.cuda = self.dev.type == "cuda"
        self.stream = torch.cuda.Stream(device=dev) if world > 1 and self.cuda else None
        self.done = None
        self.pairs = []
        self.waits = []            # (before, after) events on the MAIN stream around its wait for the collective
        self.host_ms = []          # CPU tensors (the gloo tests): the collective is synchronous, timed on the host
        self.timing = False

    def wait_previous(self):
        """Before the buffer is written again (the next backward): the previous step's collective has read it.
        In the instrumented pass the wait is bracketed by two events on the main stream: their distance is the time
        the main stream actually stood still for the collective (allreduce_exposed_ms_per_step)."""
        if self.done is not None:
            main = torch.cuda.current_stream(self.dev)
            if self.timing:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(main)
                main.wait_event(self.done)
                b.record(main)
                self.waits.append((a, b))
            else:
                main.wait_event(self.done)

    def launch(self, fused):
        if self.world == 1:
            return
        if not self.cuda:
            t0 = time.perf_counter()
            distributed.allreduce_weight_grads(fused)
            if self.timing:
                self.host_ms.append((time.perf_counter() - t0) * 1e3)
            return
        main = torch.cuda.current_stream(self.dev)
        self.stream.wait_stream(main)
        with torch.cuda.stream(self.stream):
            if self.timing:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(self.stream)
            distributed.allreduce_weight_grads(fused)
            if self.timing:
                b.record(self.stream)
                self.pairs.append((a, b))
            self.done = torch.cuda.Event()
            self.done.record(self.stream)

    def finish(self):
        if self.stream is not None:
            torch.cuda.current_stream(self.dev).wait_stream(self.stream)

    def ms_per_step(self, steps):
        if self.host_ms:
            return sum(self.host_ms) / steps
        if not self.pairs:
            return None
        return sum(a.elapsed_time(b) for a, b in self.pairs) / steps

    def exposed_ms_per_step(self, steps):
        """Time the main stream waited for the collective (0 when it hides entirely under the next step's forward)."""
        if self.host_ms:
            return sum(self.host_ms) / steps          # synchronous collective: all of it is exposed
        if not self.waits:
            return None
        return sum(a.elapsed_time(b) for a, b in self.waits) / steps


def rank_spread(local_s, steps, dev):
    """ms_per_step of the slowest and the fastest rank (the headline uses the slowest, as the contract says)."""
    lo = -distributed.max_over_ranks(-local_s, dev)
    hi = distributed.max_over_ranks(local_s, dev)
    return {"min": round(lo / steps * 1e3, 4), "max": round(hi / steps * 1e3, 4)}


# ------------------------------------------------------------------------------------------- CPU legs
def _oracle_pass(fwd, bwd, P, X, ups, filters, layers):
    acts, x = [], X
    for li in range(4):
        s = layers[li][2]
        x = stack.selu_numpy(fwd(P, x, filters[li], (s, s, s)))
        acts.append(x)
    carry, dws = None, [None] * 4
    for li in (3, 2, 1, 0):
        s = layers[li][2]
        g = ups[li] if carry is None else ups[li] + carry
        g = stack.selu_grad_numpy(acts[li], g)
        x_in = acts[li - 1] if li > 0 else X
        carry, dws[li] = bwd(g, P, x_in, filters[li], (s, s, s))
    return acts, carry, dws


def cpu_baseline(points_np, feats_np, st, ups_np, budget_s=12.0):
    """Time the reference CPU loops on the same stack; also returns the serial oracle's outputs for parity."""
    from oracle import oracle                      # checker only; never on the product path
    filters = [f.detach().cpu().numpy() for f in st.filters]
    have_ref = oracle.ref_compute("atrous_omp") is not None
    if have_ref:
        # Conv3pGradOp forces hardware_concurrency() threads (.cpp:611-619) and both ops parallelise over the batch
        # only (.cpp:453-456, :620-622): at most B threads ever have work
        cores = max(1, min(points_np.shape[0], oracle.reference_threads()))
        fwd = lambda P, x, w, s: oracle.reference_forward(P, x, w, s, stack.VOXEL, kind="atrous_omp")
        bwd = lambda g, P, x, w, s: oracle.reference_backward(g, P, x, w, s, stack.VOXEL, kind="atrous_omp")
        kind = "reference"
    else:
        cores = max(1, min(points_np.shape[0], os.cpu_count() or 1))
        fwd = lambda P, x, w, s: oracle.forward(P, x, w, s, stack.VOXEL, nthreads=cores)
        bwd = lambda g, P, x, w, s: oracle.backward(g, P, x, w, s, stack.VOXEL, nthreads=cores)
        kind = "port"
    reps, t_total = 0, 0.0
    while t_total < budget_s and reps < 50:
        t0 = time.perf_counter()
        _oracle_pass(fwd, bwd, points_np, feats_np, ups_np, filters, st.layers)
        t_total += time.perf_counter() - t0
        reps += 1
    pts = points_np.shape[0] * points_np.shape[1]
    value = pts * reps / t_total / 1e6
    # serial (deterministic) oracle on 2 clouds for the parity numbers
    sf = lambda P, x, w, s: oracle.forward(P, x, w, s, stack.VOXEL)
    sb = lambda g, P, x, w, s: oracle.backward(g, P, x, w, s, stack.VOXEL)
    ref = _oracle_pass(sf, sb, points_np[:2], feats_np[:2], [u[:2] for u in ups_np], filters, st.layers)
    what = ("the reference's own batch loops (tf_conv3p_atrous.cpp:451-504, :608-716 compiled in place, "
            "-O3 -fopenmp -DCONV_OPENMP)") if have_ref else "the oracle's C restatement"
    base = {"value": round(value, 4), "unit": "Mpoints/s", "cores": cores, "kind": kind,
            "host_threads": oracle.reference_threads() if have_ref else cores,
            "sample": "%d repetitions of the full workload (B=%d, N=%d, 4-layer stack fwd+bwd) through %s, OpenMP "
                      "over the batch, %.1f s of CPU time" % (reps, points_np.shape[0], points_np.shape[1], what,
                                                              t_total)}
    # BASELINE config 1: one cloud, first layer, forward only, serial
    P1, W1 = points_np[:1], filters[0]
    serial_ref = oracle.ref_compute("atrous") is not None
    f1 = (lambda: oracle.reference_forward(P1, P1, W1, (1, 1, 1), stack.VOXEL)) if serial_ref else \
        (lambda: oracle.forward(P1, P1, W1, (1, 1, 1), stack.VOXEL))
    f1()
    best, n1, t1 = 1e9, 0, 0.0
    while t1 < 1.5 and n1 < 200:
        t0 = time.perf_counter()
        f1()
        dt = time.perf_counter() - t0
        best, n1, t1 = min(best, dt), n1 + 1, t1 + dt
    cfg1 = {"value": round(points_np.shape[1] / best / 1e6, 4), "unit": "Mpoints/s", "ms": round(best * 1e3, 4),
            "cores": 1, "kind": "reference" if serial_ref else "port",
            "sample": "BASELINE config 1: B=1, N=2048, 3->9, stride 1, forward only, serial; best of %d calls" % n1}
    return base, cfg1, ref


def op_boundary_native(Ps):
    """The same op-by-op step WITHOUT Python: integration/op_boundary_bench (C++, links the library's C ABI only) issues
    the 8 *_cached_* calls + SELU ops on one stream over the same clouds, HIP-event timed.  Run as a subprocess, outside
    every timed region of this file."""
    exe = os.path.join(ROOT, "integration", "op_boundary_bench")
    if not os.path.exists(exe):
        return {"op_boundary_native_ms_per_step": None, "op_boundary_native_note": "integration/op_boundary_bench not built (__graft_entry__.build())"}
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".bin", delete=False) as f:
        np.asarray([len(Ps), Ps[0].shape[0], Ps[0].shape[1]], dtype=np.int32).tofile(f)
        for p in Ps:
            np.ascontiguousarray(p, dtype=np.float32).tofile(f)
        path = f.name
    try:
        out = {}
        for mode, key in (("hinted", "op_boundary_native"), ("unhinted", "op_boundary_native_unhinted")):
            r = subprocess.run([exe, "50", "10", path] + (["unhinted"] if mode == "unhinted" else []), capture_output=True, text=True, timeout=120)
            d = json.loads(r.stdout.strip().splitlines()[-1])
            out[key + "_ms_per_step"] = d["op_boundary_native_ms_per_step"]
            out[key + "_value"] = round(Ps[0].shape[0] * Ps[0].shape[1] / d["op_boundary_native_ms_per_step"] / 1e3, 3)
            out[key + "_host_enqueue_ms_per_step"] = d["host_enqueue_ms_per_step"]
        out["op_boundary_native_note"] = ("what integration/tf_conv3p_shim.cc gives an unchanged TF caller: the shim holds a reference to the "
                                          "points tensor it validated last and passes CONV3P_CACHE_POINTS_UNCHANGED for calls on that very "
                                          "buffer (first op of a step: content hash + rebuild); _unhinted: every call hashed (rounds 3-5)")
        return out
    except Exception as e:   # noqa: BLE001 -- a reported figure, never a reason to lose the bench line
        return {"op_boundary_native_ms_per_step": None, "op_boundary_native_note": "failed: %r" % (e,)}
    finally:
        os.unlink(path)


def op_boundary_cached_step(st, cache, P, X, gcat):
    """The cfg2 step as 8 independent op calls (+ SELU ops) against ONE persistent cache: no POINTS_UNCHANGED hints,
    no prefetch, no stack-level entry point -- the device-side validation alone decides what is rebuilt."""
    acts, x = [], X
    for li in range(4):
        s_ = st.layers[li][2]
        x = op.selu(op.conv3p(P, x, st.filters[li], (s_, s_, s_), stack.VOXEL, cache=cache))
        acts.append(x)
    H = stack.HIDDEN
    carry = None
    for li in (3, 2, 1, 0):
        s_ = st.layers[li][2]
        ext = gcat[:, :, H * li:H * (li + 1)].contiguous()
        g = op.selu_grad(acts[li], ext, carry)
        x_in = acts[li - 1] if li > 0 else X
        carry, _ = op.conv3p_grad(g, P, x_in, st.filters[li], (s_, s_, s_), stack.VOXEL, grad_filter_out=st.grad_views[li],
                                  cache=cache)
    return carry


# ------------------------------------------------------------------------------------------- other configs
def _rel(got, ref):
    ref = np.asarray(ref)
    return float(np.abs(np.asarray(got, dtype=np.float64) - ref).max() / max(1.0, float(np.abs(ref).max())))


def cfg4_parity(st, P, X, up, ncls):
    """One cloud of the cfg4 workload through the whole 5-layer stack, outside every timed region: the HIP path against
    the oracle's fp32 loops (ref32) and against the exact sums over the oracle's OWN pair lists, float64 accumulation
    through all layers (tests/parity_util.exact_from_oracle_lists).  Checker only."""
    from oracle import oracle
    from tests.parity_util import exact_from_oracle_lists
    t0 = time.perf_counter()
    dev = P.device
    acts = st.forward(P[:1].contiguous(), X[:1].contiguous())
    dx, fused = st.backward([up[:1].contiguous()])
    torch.cuda.synchronize(dev)
    acts = [a.cpu().numpy()[0] for a in acts]
    dx, fused = dx.cpu().numpy()[0], fused.cpu().numpy().copy()
    Pn, Xn, upn = P[:1].cpu().numpy(), X[:1].cpu().numpy(), up[:1].cpu().numpy()
    filters = [f.detach().cpu().numpy() for f in st.filters]
    H = stack.HIDDEN

    def run(conv_f, conv_b, dt):
        a, x = [], Xn[0].astype(dt)
        for li in range(4):
            s_ = st.layers[li][2]
            x = stack.selu_numpy(conv_f(x, filters[li], (s_, s_, s_)))
            a.append(x)
        concat = np.concatenate(a, axis=1)
        logits = stack.selu_numpy(conv_f(concat, filters[4], (1, 1, 1)))
        a.append(logits)
        g = stack.selu_grad_numpy(logits, upn[0].astype(dt))
        dws = [None] * 5
        dconcat, dws[4] = conv_b(g, concat, filters[4], (1, 1, 1))
        carry = None
        for li in (3, 2, 1, 0):
            s_ = st.layers[li][2]
            e = dconcat[:, H * li:H * (li + 1)]
            g = stack.selu_grad_numpy(a[li], e if carry is None else e + carry)
            carry, dws[li] = conv_b(g, a[li - 1] if li > 0 else Xn[0].astype(dt), filters[li], (s_, s_, s_))
        return a, carry, np.concatenate([w.reshape(-1) for w in dws])

    r32 = run(lambda x, w, s_: oracle.forward(Pn, x[None], w, s_, stack.VOXEL)[0],
              lambda g, x, w, s_: tuple(v[0] if i == 0 else v for i, v in enumerate(oracle.backward(g[None], Pn, x[None], w, s_, stack.VOXEL))),
              np.float32)
    ex = run(lambda x, w, s_: exact_from_oracle_lists(Pn[0], x, w, np.zeros((x.shape[0], w.shape[-1])), s_, stack.VOXEL)[0],
             lambda g, x, w, s_: exact_from_oracle_lists(Pn[0], x, w, g, s_, stack.VOXEL)[1:],
             np.float64)
    hip = (acts, dx, fused)
    cmp_ = lambda u, v: {"y": max(_rel(a, b) for a, b in zip(u[0], v[0])), "dX": _rel(u[1], v[1]), "dW": _rel(u[2], v[2])}
    return {"sample": "cloud 0 of the workload's first batch, all five layers; max |delta| / max(1, max |ref|); exact = float64 sums over "
                      "the oracle's own (fp32-decided) pair lists, chained through the layers",
            "hip_vs_ref32": cmp_(hip, r32), "hip_vs_exact": cmp_(hip, ex), "ref32_vs_exact": cmp_(r32, ex),
            "cpu_seconds": round(time.perf_counter() - t0, 1)}


def cfg5_parity(P, X, W, dY):
    """One cloud of the cfg5 shard (128 -> 256, sums of ~10^4 terms), outside every timed region: every channel against the
    exact sums over the oracle's pair lists; the oracle's own fp32 loops on channel slices (16 output channels of y, 8 input
    channels of dX / dW: the reference's loops make every channel an independent sum).  Checker only."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    from tests.parity_util import exact_from_oracle_lists
    t0 = time.perf_counter()
    s_ = (1, 1, 1)
    p1, x1, dy1 = P[:1].contiguous(), X[:1].contiguous(), dY[:1].contiguous()
    y = op.conv3p(p1, x1, W, s_, stack.VOXEL).cpu().numpy()[0]
    dx, dw = op.conv3p_grad(dy1, p1, x1, W, s_, stack.VOXEL)
    dx, dw = dx.cpu().numpy()[0], dw.cpu().numpy()
    Pn, Xn, Wn, dYn = p1.cpu().numpy(), x1.cpu().numpy(), W.cpu().numpy(), dy1.cpu().numpy()
    cs, ks = slice(0, 16), slice(0, 8)
    with ThreadPoolExecutor(max_workers=3) as ex:
        f_y = ex.submit(lambda: oracle.forward(Pn, Xn, np.ascontiguousarray(Wn[..., cs]), s_, stack.VOXEL)[0])
        f_b = ex.submit(lambda: oracle.backward(dYn, Pn, np.ascontiguousarray(Xn[:, :, ks]), np.ascontiguousarray(Wn[:, :, :, ks, :]), s_, stack.VOXEL))
        ye, dxe, dwe = exact_from_oracle_lists(Pn[0], Xn[0], Wn, dYn[0], s_, stack.VOXEL)
        y32, (dx32, dw32) = f_y.result(), f_b.result()
    dx32 = dx32[0]
    # (relative to the WHOLE tensor's maximum, as the tests do: rel_err of a slice against the slice's own maximum would
    # be a different, larger figure)
    rel = lambda got, ref, full: float(np.abs(np.asarray(got, np.float64) - ref).max() / max(1.0, float(np.abs(full).max())))
    return {"sample": "cloud 0 of the shard's first batch; max |delta| / max(1, max |tensor|); hip_vs_exact over every channel; the fp32 reference "
                      "loops on y[:, 0:16], dX[:, 0:8], dW[..., 0:8, :]; exact = float64 sums over the oracle's own pair lists",
            "hip_vs_ref32": {"y": rel(y[:, cs], y32, ye), "dX": rel(dx[:, ks], dx32, dxe), "dW": rel(dw[:, :, :, ks, :], dw32, dwe)},
            "hip_vs_exact": {"y": rel(y, ye, ye), "dX": rel(dx, dxe, dxe), "dW": rel(dw, dwe, dwe)},
            "ref32_vs_exact": {"y": rel(y32, ye[:, cs], ye), "dX": rel(dx32, dxe[:, ks], dxe), "dW": rel(dw32, dwe[:, :, :, ks, :], dwe)},
            "cpu_seconds": round(time.perf_counter() - t0, 1)}



def cfg4_report(lib, dev, steps=20, warmup=5, parity=False):
    """S3DIS scene_seg stack (pointcnn_scene_seg_acsd.py:51-57): B=16, N=4096, C_in=9, 13 classes."""
    B, N, cin, ncls = 16, 4096, 9, 13
    Ps = [torch.from_numpy(synth.room_like(B, N, 40 + i)).to(dev) for i in range(3)]
    Xs = [torch.from_numpy(synth.features(B, N, cin, 50 + i, points=p.cpu().numpy())).to(dev) for i, p in enumerate(Ps)]
    st = stack.Conv3pStack(cin, ncls, device=dev, seed=3)
    st.tune(Ps[0])
    ups = [torch.from_numpy(synth.upstream_grad(B, N, ncls, 60)).to(dev)]
    ctr = [0]

    def step():
        i = ctr[0] % 3
        ctr[0] += 1
        st.forward(Ps[i], Xs[i])
        st.prefetch(Ps[(i + 1) % 3])
        st.backward(ups)

    dt = timed(dev, step, steps, warmup)
    bpp = stack_bytes_per_point(st.layers)
    achieved = bpp * B * N / dt / 1e9
    kinds = profile_steps(lib, dev, step, 5)
    out = {"workload": "cfg4 S3DIS scene_seg-shaped: B=16 x N=4096 room blocks, C_in=9, conv3p stack 9->9 s1..s4 + 36->13 "
                       "s1 (+SELU), forward+backward, a different batch every step",
           "ms_per_step": round(dt * 1e3, 4), "value": round(B * N / dt / 1e6, 3), "unit": "Mpoints/s",
           "roofline": {"bound": "hbm", "scope": "whole step", "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "bytes_per_point": bpp},
           "kernel_ms_per_step": {k: round(v[1] / 5, 4) for k, v in kinds.items()}}
    if parity:
        try:
            out["parity"] = cfg4_parity(st, Ps[0], Xs[0], ups[0], ncls)
        except Exception as e:   # noqa: BLE001 -- a reported figure, never a reason to lose the bench line
            out["parity"] = {"failed": repr(e)}
    del st
    return out


NB_CFG5 = 3     # distinct batches cycled by the cfg5 legs: odd, so that neither of the two caches ever re-meets its content


class GeometryPrefetch:
    """Two neighbour caches used in turn: while a step runs on the current stream, the NEXT batch's geometry (sort +
    search of its clouds) is built in the other cache on a side stream -- what Conv3pStack.prefetch does for the models'
    stacks, for a single layer through the op-level API (conv3p_cache_prepare_*)."""

    def __init__(self, dev, B, N, ci, co, enabled=True):
        self.dev, self.enabled = dev, enabled
        n = 2 if enabled else 1
        self.caches = [op.NeighborCache(B, N, torch.float32, dev, slots=1, max_taps=27, max_cin=ci, max_cout=co) for _ in range(n)]
        self.side = torch.cuda.Stream(device=dev) if enabled else None
        self.ready = [torch.cuda.Event() for _ in range(n)]
        self.done = [torch.cuda.Event() for _ in range(n)]
        self.pending = [None] * n       # the points tensor whose geometry cache k holds (or is being given)
        self.k = 0

    def cache_for(self, P):
        """Cache holding the geometry of P (built here if nobody prefetched it) and whether the hint applies."""
        if not self.enabled:
            return self.caches[0], False
        main = torch.cuda.current_stream(self.dev)
        k = self.k
        if self.pending[k] is not P:    # first step: build it in line
            op.cache_prepare(P, (3, 3, 3), (1, 1, 1), stack.VOXEL, self.caches[k], deep_orders=True)
            self.pending[k] = P
        else:
            main.wait_event(self.ready[k])
        return self.caches[k], True

    def prefetch(self, P_next):
        """Call after the forward has been enqueued: the other cache was last used by the previous step."""
        if not self.enabled:
            return
        o = self.k ^ 1
        self.side.wait_event(self.done[o])
        with torch.cuda.stream(self.side):
            op.cache_prepare(P_next, (3, 3, 3), (1, 1, 1), stack.VOXEL, self.caches[o], stream=self.side, deep_orders=True)
            self.ready[o].record(self.side)
        self.pending[o] = P_next

    def step_done(self):
        if not self.enabled:
            return
        self.done[self.k].record(torch.cuda.current_stream(self.dev))
        self.k ^= 1


def cfg5_report(lib, dev, steps=5, warmup=2, parity=False):
    """cfg5 per-GPU shard: B=16, N=8192 SceneNN-shaped rooms, ONE 128->256 layer, stride 1, forward+backward,
    geometry rebuilt every step (three cycled batches over the two caches)."""
    B, N, ci, co = 16, 8192, 128, 256
    t = lambda a: torch.from_numpy(a).to(dev)
    # THREE batches over the two alternating caches: a cache never meets the content it already holds (with two, cache k
    # would always be handed batch k, its hash would match and nothing would be rebuilt -- ADVICE r3)
    Ps = [t(synth.room_like(B, N, 7 + i, extent=(2.4, 2.4, 3.0))) for i in range(NB_CFG5)]
    X = t(synth.features(B, N, ci, 8, points=Ps[0].cpu().numpy()))
    dY = t(synth.upstream_grad(B, N, co, 9))
    W = t(synth.filter_weights(3, 3, 3, ci, co, 5))
    def run(prefetch):
        geo = GeometryPrefetch(dev, B, N, ci, co, enabled=prefetch)
        ctr = [0]

        def step():
            P, Pn = Ps[ctr[0] % NB_CFG5], Ps[(ctr[0] + 1) % NB_CFG5]
            ctr[0] += 1
            cache, hint = geo.cache_for(P)
            op.conv3p(P, X, W, (1, 1, 1), stack.VOXEL, cache=cache, points_unchanged=hint)
            geo.prefetch(Pn)
            op.conv3p_grad(dY, P, X, W, (1, 1, 1), stack.VOXEL, cache=cache, points_unchanged=True)
            geo.step_done()

        dt = timed(dev, step, steps, warmup)
        kinds = profile_steps(lib, dev, step, 2)
        torch.cuda.synchronize(dev)
        del geo
        return dt, kinds

    dt_serial, _ = run(False)
    dt, kinds = run(True)
    useful = 3 * 2 * 27 * ci * co * B * N            # fwd + dX + dW, dense-equivalent
    achieved = useful / dt / 1e12
    # the matrix instructions actually ISSUED per step (SQ_INSTS_MFMA of a separate rocprofv3 --pmc pass over the same
    # workload, profiles/deep_mfma_latest.json): only populated (tile, tap) products are issued
    mf, mf_stale = load_counters("deep_mfma")
    issued = None if mf is None else mf["mfma_instructions_per_step"] * mf["flops_per_instruction"] / dt / 1e12
    par = None
    if parity:
        try:
            par = cfg5_parity(Ps[0], X, W, dY)
        except Exception as e:   # noqa: BLE001
            par = {"failed": repr(e)}
    return {"parity": par, "workload": "cfg5 per-GPU shard: B=16 x N=8192 SceneNN-shaped rooms, one conv3p layer 128->256, stride 1, "
                        "forward+backward, a different batch every step, the next batch's geometry (sort + search) built in a "
                        "second cache on a side stream during the step (as the headline does); ms_per_step_geometry_in_line: "
                        "the same with the geometry built in line",
            "ms_per_step": round(dt * 1e3, 3), "ms_per_step_geometry_in_line": round(dt_serial * 1e3, 3),
            "value": round(B * N / dt / 1e6, 3), "unit": "Mpoints/s",
            "roofline": {"bound": "mfma", "scope": "whole step", "achieved": round(achieved, 2),
                         "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4),
                         "achieved_issued": None if issued is None else round(issued, 2),
                         "frac_issued": None if issued is None else round(issued / MFMA_F32_PEAK_TFLOPS, 4),
                         "counters_stale": bool(mf_stale),
                         "flops_per_point": 3 * 2 * 27 * ci * co,
                         "note": "useful (dense-equivalent) flops = 3 x 2*27*Cin*Cout per point; the matrix "
                                 "instructions ISSUED are fewer (only populated (tile, tap) products run): "
                                 "achieved_issued / frac_issued = SQ_INSTS_MFMA x 4096 flops over the same step time"},
            "kernel_ms_per_step": {k: round(v[1] / 2, 4) for k, v in kinds.items()},
            "kernel_ms_per_step_note": "HIP events around each launch in the prefetch leg: prep / search / deep_order run on "
                                       "the side stream beside the main stream's kernels, so their figures are overlapped "
                                       "LATENCIES, not kernel cost (alone: profiles/r05_deep_kernel_stats.txt, r05_geometry_time.txt)"}


def head_report(lib, dev, steps=20, warmup=3):
    """The classification model's dense head at model size (pointcnn2_acsd.py:69-75): fc1 73 728 x 512 + fc2, forward
    and backward, B=32 -- bound by one read (forward) / one read + one write (backward) of the 151 MB fc1 matrix."""
    from pointwise_amd import head
    B, N = B_PER_GPU, N_POINTS
    hd = head.ClassificationHead(N, num_class=40, device=dev, seed=5)
    feat = torch.randn((B, N, 36), device=dev)
    labels = torch.randint(0, 40, (B,), device=dev)
    mask = (torch.rand((B, 512), device=dev) < 0.5).float()
    def fwd():
        return hd.forward(feat, training=True, keep_mask=mask)
    def both():
        logits = fwd()
        _, dlogits = hd.loss(logits, labels)
        hd.backward(dlogits)
    tf_ = timed(dev, fwd, steps, warmup)
    tb_ = timed(dev, both, steps, warmup)
    wbytes = hd.W1.numel() * 4
    out = {"workload": "classification head: view (32, 73728) -> fc 512 selu -> dropout_selu -> fc 40 selu, softmax "
                       "cross-entropy, forward+backward (fc1 weights 151 MB)",
           "forward_ms": round(tf_ * 1e3, 4), "forward_backward_ms": round(tb_ * 1e3, 4),
           "roofline": {"bound": "hbm", "scope": "forward+backward", "achieved": round(3 * wbytes / tb_ / 1e9, 1),
                        "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(3 * wbytes / tb_ / 1e9 / HBM_PEAK_GBS, 4),
                        "bytes": 3 * wbytes, "note": "algorithmic bytes = fc1 read (forward) + fc1 read + dW1 write (backward)"}}
    del hd
    return out


# ------------------------------------------------------------------------------------------- launch plumbing
def respawn_under_torchrun(args):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE: run the same command one rank per GPU."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline legs")
    ap.add_argument("--no-extra", action="store_true", help="skip other_configs / stateless / isolated passes")
    ap.add_argument("--serial", action="store_true",
                    help="developer: no side stream at all (clean per-kernel durations under rocprofv3)")
    ap.add_argument("--workload", choices=("cfg2", "cfg5"), default="cfg2",
                    help="cfg2 (default; cfg3 at --gpus 8): B=32 x N=2048 per GPU, 4-layer stack.  cfg5: B=16 x N=8192 per "
                         "GPU, one 128->256 layer, one 3.54 MB all-reduce")
    ap.add_argument("--fused", choices=("forward", "backward"), default=None, help="developer: only that pass fused")
    ap.add_argument("--no-fused", action="store_true", help="developer A/B: per-layer launches only (the default lets "
                    "Conv3pStack.tune() fuse the forward pass's hidden layers into one launch for clouds with short pair lists)")
    ap.add_argument("--fused-stack", action="store_true",
                    help="developer: the hidden layers of a pass as ONE launch (CONV3P_CACHE_FUSED_STACK, opt-in; echoed "
                         "into config; the default run reports it beside the headline as fused_stack_ms_per_step)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="do not enqueue the next batch's neighbour search under the current batch's backward")
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (the product path has no CPU fallback)")
    if torch.cuda.device_count() < args.gpus:
        raise SystemExit("--gpus %d but only %d HIP device(s) visible" % (args.gpus, torch.cuda.device_count()))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_torchrun(args)
    rank, world, local = distributed.init_from_env()
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    lib = _lib.load()
    if args.workload == "cfg5":
        return main_cfg5(args, lib, dev, rank, world)

    # synthetic ModelNet40-shaped batches; features == points (modelnet_provider.py:212-213).
    # Every step gets a DIFFERENT batch (NBATCH distinct batches resident in HBM, cycled), as in training:
    # the neighbour cache therefore rebuilds its geometry every step (1 sort + 4 searches per step) and only
    # saves the repeats INSIDE a step (8 op calls share one `points`).  Nothing is carried across steps.
    NBATCH = 4
    Ps = [synth.modelnet_like(B_PER_GPU, N_POINTS, seed=1234 + 2 + 1000 * rank + 17 * i) for i in range(NBATCH)]
    if os.environ.get("CONV3P_BENCH_PRESORT"):   # developer experiment: clouds arrive in Morton order (gather locality)
        def morton(P):
            q = np.clip(((P - P.min(axis=1, keepdims=True)) / 0.05).astype(np.int64), 0, 1023)
            code = np.zeros(P.shape[:2], dtype=np.int64)
            for bit in range(10):
                for a in range(3):
                    code |= ((q[..., a] >> bit) & 1) << (3 * bit + a)
            return np.argsort(code, axis=1, kind="stable")
        Ps = [np.take_along_axis(P, morton(P)[..., None], axis=1) for P in Ps]
    P = Ps[0]
    ups_np = [synth.upstream_grad(B_PER_GPU, N_POINTS, stack.HIDDEN, 77 + li + 1000 * rank) for li in range(4)]
    tPs = [torch.from_numpy(p).to(dev) for p in Ps]
    tXs = [t.clone() for t in tPs]
    tP, tX = tPs[0], tXs[0]
    ups = [torch.from_numpy(u).to(dev) for u in ups_np]
    gcat = torch.cat(ups, dim=2).contiguous()      # dL/dconcat (B, N, 36): what the model's dense head hands back
    st = stack.Conv3pStack(C_IN, None, device=dev, seed=1234, overlap_search=not args.serial, fused_launch=True if args.fused_stack else (args.fused or (False if args.no_fused else "auto")))
    counter = [0]
    # set-up, not a step: create the RCCL communicator (seconds on the first collective), allocate the stack's
    # neighbour caches and load the library's code object (one 64-point call) before anything is timed,
    # whatever --warmup is
    distributed.allreduce_weight_grads(torch.zeros_like(st.fused_grad))
    distributed.barrier()
    st.tune(tPs[0])                 # set-up: short or long pair lists on these clouds? -> one of the two cache hints
                                    # (without it the library decides on the device: one empty launch per dilated layer)
    st.prepare(B_PER_GPU, N_POINTS)
    _p = tPs[0][:1, :64].contiguous()
    op.conv3p(_p, _p, st.filters[0], (1, 1, 1), stack.VOXEL)
    torch.cuda.synchronize(dev)

    prefetch = not (args.no_prefetch or args.serial)
    rccl_world = torch.distributed.get_world_size() if world > 1 else 1     # after the first collective (above)
    red = Reducer(dev, world)

    def make_step(stk, pre):
        def step():
            i = counter[0] % NBATCH
            counter[0] += 1
            stk.forward(tPs[i], tXs[i])
            if pre:
                # the next batch's geometry (1 sort + 4 searches) goes to the side stream now and runs under this
                # batch's backward, as a data-loader-fed training loop would do.  Every step still performs exactly
                # one full geometry build, one forward and one backward inside the timed region.
                stk.prefetch(tPs[(i + 1) % NBATCH])
            red.wait_previous()                 # the buffer the backward writes: the last all-reduce has consumed it
            dx, fused = stk.backward(gcat)
            red.launch(fused)                   # N > 1: on the communication stream, under the next step's forward
            return dx, fused
        return step

    step = make_step(st, prefetch)

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
    red.finish()
    torch.cuda.synchronize(dev)
    local_elapsed = time.perf_counter() - t0     # this rank's own clock, before the closing barrier
    distributed.barrier()
    elapsed = distributed.max_over_ranks(time.perf_counter() - t0, dev)
    spread = rank_spread(local_elapsed, args.steps, dev) if world > 1 else None

    # ---- per-kernel HIP-event timing of the same K steps (instrumented, not the timed region) ----
    red.timing = True
    kinds = profile_steps(lib, dev, step, args.steps)
    red.finish()
    torch.cuda.synchronize(dev)
    red.timing = False
    allreduce_ms = red.ms_per_step(args.steps)
    allreduce_exposed_ms = red.exposed_ms_per_step(args.steps)
    extra = world == 1 and not args.no_extra
    kinds_iso = None
    if extra and prefetch:
        # the same steps with no side stream at all: kernels run alone, one after the other
        st_iso = stack.Conv3pStack(C_IN, None, device=dev, seed=1234, overlap_search=False, fused_launch=st.fused_launch)
        st_iso.sparse_neighbourhoods = st.sparse_neighbourhoods
        st_iso.prepare(B_PER_GPU, N_POINTS)
        step_iso = make_step(st_iso, False)
        for _ in range(3):
            step_iso()
        kinds_iso = profile_steps(lib, dev, step_iso, min(args.steps, 20))
        del st_iso

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
            ttab, stale_t = load_counters("traffic")
            _, stale_v = load_counters("valu")
            counters_stale = stale_t or stale_v
            if ttab is not None:
                try:
                    t = ttab.get(dom)
                    # measured in separate rocprofv3 --pmc passes (tools/collect_profiles.sh): HBM bytes per launch,
                    # FETCH_SIZE raw + WRITE_SIZE (lower bound; see tools/traffic_json.py for the gfx950 caveat)
                    traffic = t["hbm_bytes_lower"] if isinstance(t, dict) else t
                except Exception:
                    traffic = None
            roofline = {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic,
                        "avg_launch_us": round(avg_s * 1e6, 2), "bytes_per_launch": int(bytes_per_launch),
                        "avg_launch_us_note": "measured with the headline's side-stream overlap (other kernels share "
                                              "the CUs); achieved/frac use this figure; prep / search entries of "
                                              "kernel_ms_per_step are side-stream latencies under that overlap "
                                              "(kernel_ms_per_step_isolated: alone)",
                        "kernel_ms_per_step": {k: round(v[1] / args.steps, 4) for k, v in kinds.items()}}
            if counters_stale:
                # the counter files were collected for other kernel sources: traffic / valu_issue_frac are omitted
                roofline["counters_stale"] = True
            vf = valu_issue_fracs(kinds, args.steps)
            if vf:
                roofline["valu_issue_frac"] = vf
                roofline["valu_issue_note"] = ("SQ_INSTS_VALU per launch (rocprofv3 --pmc pass of this bench, "
                                               "profiles/valu_latest.json) / (1024 SIMDs x 2.4 GHz / 4 x launch duration "
                                               "measured now); cfg2 is cache-resident, so this -- not HBM -- is the "
                                               "resource its kernels load")
            if kinds_iso and dom in kinds_iso:
                ni, msi = kinds_iso[dom]
                roofline["avg_launch_us_isolated"] = round(msi / ni * 1e3, 2)
                roofline["frac_isolated"] = round(bytes_per_launch / (msi / ni * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)
                nst = min(args.steps, 20)
                roofline["kernel_ms_per_step_isolated"] = {k: round(v[1] / nst, 4) for k, v in kinds_iso.items()}
            bpp = stack_bytes_per_point(st.layers)
            roofline["whole_step"] = {"bytes_per_point": bpp,
                                      "achieved": round(bpp * total_pts / world / (elapsed / args.steps) / 1e9, 2),
                                      "frac": round(bpp * total_pts / world / (elapsed / args.steps) / 1e9 / HBM_PEAK_GBS, 5)}
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
                          "parallelism": "dp%d" % world,
                          "backward_kernel_choice": "hint from Conv3pStack.tune() at set-up: " +
                                                    ("short lists" if st.sparse_neighbourhoods else "long lists")
                                                    if st.sparse_neighbourhoods is not None else "on the device"},
               "roofline": roofline}
        if os.environ.get("CONV3P_HIP_LIB"):   # developer A/B builds: never silently
            out["config"]["hip_library_override"] = os.environ["CONV3P_HIP_LIB"]
        if os.environ.get("CONV3P_BENCH_PRESORT"):   # (the input order of the clouds is part of the workload: never silently)
            out["config"]["developer_presorted_clouds"] = "Morton order (CONV3P_BENCH_PRESORT): NOT the headline workload"
        out["config"]["fused_launches"] = {"auto": "none (tune() saw long pair lists)", False: "none (--no-fused)", True: "forward and backward (--fused-stack)",
                                           "forward": "the forward pass's hidden layers as ONE launch (CONV3P_CACHE_FUSED_FORWARD, set by Conv3pStack.tune() for short pair lists)",
                                           "backward": "backward only (--fused backward)"}[st.fused_launch]
        out["config"]["fused_status"] = list(st.fused_status())
        if world > 1:
            out["rccl_world"] = rccl_world
            out["allreduce_ms_per_step"] = None if allreduce_ms is None else round(allreduce_ms, 4)
            out["allreduce_exposed_ms_per_step"] = None if allreduce_exposed_ms is None else round(allreduce_exposed_ms, 4)
            out["ms_per_step_ranks"] = spread
            out["allreduce_bytes"] = int(st.fused_grad.numel() * 4)
        if extra and not (args.fused_stack or args.fused or args.no_fused):
            # the same step with per-layer launches only / with BOTH passes' hidden layers fused
            for key, fl in (("per_layer_launches_ms_per_step", False), ("fused_both_passes_ms_per_step", True)):
                st_f = stack.Conv3pStack(C_IN, None, device=dev, seed=1234, overlap_search=not args.serial, fused_launch=fl)
                st_f.sparse_neighbourhoods = st.sparse_neighbourhoods
                st_f.prepare(B_PER_GPU, N_POINTS)
                step_f = make_step(st_f, prefetch)
                dt_f = timed(dev, step_f, min(args.steps, 30), 5)
                out[key] = round(dt_f * 1e3, 4)
                del st_f
        if extra:
            # the drop-in boundary as the TF shim drives it: stateless ops, SELU as separate ops
            st_plain = stack.Conv3pStack(C_IN, None, device=dev, seed=1234, use_cache=False)
            step_plain = make_step(st_plain, False)
            dt_plain = timed(dev, step_plain, min(args.steps, 20), 3)
            out["stateless_ms_per_step"] = round(dt_plain * 1e3, 4)
            out["stateless_value"] = round(total_pts / dt_plain / 1e6, 3)
            del st_plain
            # ... and through the cached entry points with a persistent cache but without any hint or prefetch
            # (_unhinted: every call validated by content hash; the default figure: the cache trusts tensor identity as the
            # TF shim does -- same storage, address and torch version counter as the tensor it validated last, reference held)
            for trust, key in ((True, "op_boundary_cached"), (False, "op_boundary_cached_unhinted")):
                cache = op.NeighborCache(B_PER_GPU, N_POINTS, torch.float32, dev, slots=4, max_taps=27, max_cin=9, max_cout=9,
                                         trust_tensor_identity=trust)
                bctr = [0]

                def step_boundary():
                    i = bctr[0] % NBATCH
                    bctr[0] += 1
                    return op_boundary_cached_step(st, cache, tPs[i], tXs[i], gcat)
                dt_b = timed(dev, step_boundary, min(args.steps, 20), 3)
                out[key + "_ms_per_step"] = round(dt_b * 1e3, 4)
                out[key + "_value"] = round(total_pts / dt_b / 1e6, 3)
                del cache
            out.update(op_boundary_native(Ps))
            out["other_configs"] = {"cfg4": cfg4_report(lib, dev, parity=not args.no_cpu), "cfg5_shard": cfg5_report(lib, dev, parity=not args.no_cpu),
                                    "classification_head": head_report(lib, dev)}
        if world == 1 and not args.no_cpu:
            base, cfg1, ref = cpu_baseline(P, P.copy(), st, ups_np)
            out["cpu_baseline"] = base
            out["cpu_baseline_cfg1"] = cfg1
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


def main_cfg5(args, lib, dev, rank, world):
    """BASELINE config 5 (per-GPU shard x world): B=16 x N=8192 SceneNN-shaped rooms per GPU, ONE 128->256 conv3p layer,
    forward+backward, then the 884 736-float (3.54 MB) sum all-reduce of grad_filter on the communication stream."""
    B, N, ci, co = 16, 8192, 128, 256
    t = lambda a: torch.from_numpy(a).to(dev)
    Ps = [t(synth.room_like(B, N, 7 + i + 1000 * rank, extent=(2.4, 2.4, 3.0))) for i in range(NB_CFG5)]
    X = t(synth.features(B, N, ci, 8 + 1000 * rank, points=Ps[0].cpu().numpy()))
    dY = t(synth.upstream_grad(B, N, co, 9 + 1000 * rank))
    W = t(synth.filter_weights(3, 3, 3, ci, co, 5))
    dW = torch.zeros_like(W)
    geo = GeometryPrefetch(dev, B, N, ci, co, enabled=not args.serial)
    distributed.allreduce_weight_grads(torch.zeros_like(dW))      # set-up: RCCL communicator
    distributed.barrier()
    rccl_world = torch.distributed.get_world_size() if world > 1 else 1
    red = Reducer(dev, world)
    ctr = [0]

    def step():
        P, Pn = Ps[ctr[0] % NB_CFG5], Ps[(ctr[0] + 1) % NB_CFG5]
        ctr[0] += 1
        cache, hint = geo.cache_for(P)
        op.conv3p(P, X, W, (1, 1, 1), stack.VOXEL, cache=cache, points_unchanged=hint)
        geo.prefetch(Pn)
        red.wait_previous()
        op.conv3p_grad(dY, P, X, W, (1, 1, 1), stack.VOXEL, grad_filter_out=dW, cache=cache, points_unchanged=True)
        red.launch(dW)
        geo.step_done()

    steps, warmup = args.steps, args.warmup
    for _ in range(warmup):
        step()
    red.finish()
    distributed.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    red.finish()
    torch.cuda.synchronize(dev)
    local_elapsed = time.perf_counter() - t0
    distributed.barrier()
    elapsed = distributed.max_over_ranks(time.perf_counter() - t0, dev)
    spread = rank_spread(local_elapsed, steps, dev) if world > 1 else None
    red.timing = True
    kinds = profile_steps(lib, dev, step, min(steps, 5))
    red.finish()
    torch.cuda.synchronize(dev)
    allreduce_ms = red.ms_per_step(min(steps, 5))
    allreduce_exposed_ms = red.exposed_ms_per_step(min(steps, 5))
    if rank == 0:
        dt = elapsed / steps
        useful = 3 * 2 * 27 * ci * co * B * N
        achieved = useful / dt / 1e12
        out = {"metric": "conv3p fwd+bwd Mpoints/s", "value": round(B * N * world / dt / 1e6, 3), "unit": "Mpoints/s",
               "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": round(dt * 1e3, 4),
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "cfg5 SceneNN-shaped: B=16 clouds/GPU x N=8192, one conv3p layer 128->256, stride 1, "
                                      "forward+backward, a different batch every step, "
                                      + ("geometry built in line" if args.serial else
                                         "the next batch's geometry built on a side stream during the step")
                                      + (", RCCL all-reduce of 884736 weight grads (3.54 MB) on a communication stream"
                                         if world > 1 else ""),
                          "global_batch": B * world, "points_per_cloud": N, "parallelism": "dp%d" % world},
               "roofline": {"bound": "mfma", "scope": "whole step, per GPU", "achieved": round(achieved, 2),
                            "peak": MFMA_F32_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / MFMA_F32_PEAK_TFLOPS, 4),
                            "flops_per_point": 3 * 2 * 27 * ci * co,
                            "kernel_ms_per_step": {k: round(v[1] / min(steps, 5), 4) for k, v in kinds.items()}}}
        if world > 1:
            out["rccl_world"] = rccl_world
            out["allreduce_ms_per_step"] = None if allreduce_ms is None else round(allreduce_ms, 4)
            out["allreduce_exposed_ms_per_step"] = None if allreduce_exposed_ms is None else round(allreduce_exposed_ms, 4)
            out["ms_per_step_ranks"] = spread
            out["allreduce_bytes"] = int(dW.numel() * 4)
        print(json.dumps(out), flush=True)
    distributed.barrier()


if __name__ == "__main__":
    main()
