"""The conv3p layer stacks of the reference's two models, driven through the operator mirror.

Classification (/root/reference/pointcnn2_acsd.py:48-67):
    Cin -> 9 (stride 1) -> 9 (stride 2) -> 9 (stride 3) -> 9 (stride 4), SELU after each layer,
    features = concat of the four activations (36 channels); all layers share `points` and voxel 0.1.
Segmentation (/root/reference/scene_seg/pointcnn_scene_seg_acsd.py:51-57):
    the same four layers, then concat(36) -> num_class (stride 1), SELU.

Only the conv3p layers and the SELU between them are here (the hot path of SURVEY.md section 8, row A8);
the dense head, loss, optimizer and data pipeline of the reference are out of scope.
forward() keeps the activations; backward() takes dL/d(activation) of every layer that has an external
consumer (the concat for classification, the logits for segmentation) and returns dL/dinput and the weight
gradients, written into ONE fused buffer so that data-parallel training needs a single all-reduce.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from . import conv3p_op as op
from . import synth

VOXEL = 0.1                       # tf.constant([0.1]), pointcnn2_acsd.py:46
CLS_STRIDES = (1, 2, 3, 4)        # pointcnn2_acsd.py:47-65
HIDDEN = 9


def _side_stream(device):
    """The stream the geometry of the next batch is built on.  (HIP stream priorities for it or for the step's own
    stream were measured twice, rounds 2 and 4: no effect -- profiles/HISTORY.md.)"""
    return torch.cuda.Stream(device=device)


class Conv3pStack:
    def __init__(self, in_channels, num_class=None, device="cuda:0", dtype=torch.float32, seed=1234,
                 use_cache=True, overlap_search=True, fuse_selu=True, c_stack=True, fused_launch="auto"):
        """num_class=None: classification stack (4 layers); an int: segmentation stack (5 layers).
        use_cache: keep the geometry (sorted points, populations, neighbour lists) of each layer's stencil in
        a NeighborCache so that the search runs once per (points, stride) instead of once per op call.
        c_stack: drive a pass with ONE call of the stack-level C entry points (conv3p_stack_forward / _backward /
        _prefetch, include/conv3p.h): one host crossing per pass, activations written straight into the concat
        buffer.  Needs use_cache and fuse_selu; shapes outside the register-resident list fall back to the op-by-op
        composition below (same results)."""
        self.use_cache = use_cache
        # fused_launch (needs c_stack): the hidden layers of a pass as ONE launch with per-cloud barriers
        # (CONV3P_CACHE_FUSED_FORWARD / _BACKWARD, csrc/conv3p_stack_fused.hpp).  True = both passes, "forward" / "backward",
        # False = never; "auto" (default) = nothing until tune() has seen the data, then the FORWARD pass for clouds with short
        # pair lists (measured: cfg2 0.405 against 0.420 ms; the fused backward keeps the next batch's search out, and on the
        # rooms of cfg4, whose tiles differ far more, both passes lose -- include/conv3p.h)
        self.fused_launch = fused_launch
        # None: the backward kernel of the dilated layers is chosen on the device from the lists themselves; True / False:
        # the CONV3P_CACHE_SPARSE / DENSE_NEIGHBOURHOODS hint for the stack's caches, set by tune() (or by hand before the
        # first batch) -- saves one empty launch per dilated layer and step
        self.sparse_neighbourhoods = None
        self.c_stack = c_stack and use_cache and fuse_selu
        self._inflight = None          # cache index of the batch between forward() and backward()
        self._pending = {}             # cache index -> points tensor whose geometry was prefetched into it
        self._scratch = None
        # SELU fused into the op's epilogues (conv3p_layer_*): 1 activation launch per step instead of 8
        self.fuse_selu = fuse_selu and use_cache
        # prefetch(): one search launch for all layers (conv3p_cache_prepare_multi_*) instead of one per layer.
        # Off by default: the batched launch is 25 % shorter on an idle GPU (170 vs 4 x 56 us for cfg2) but its
        # 4096 pending workgroups keep refilling the CUs and starve the concurrent backward kernels (78 KB of
        # LDS each) -- measured 0.54 vs 0.49 ms per step.
        self.batched_prefetch = False
        # enqueue every layer's neighbour search on a second stream at the start of forward(): the search of
        # layer l+1 (VALU-bound) then runs while layer l accumulates (gather-latency-bound)
        self.overlap_search = overlap_search and use_cache
        self._side = None
        self._cache = None
        self._caches = [None, None]    # two neighbour caches: the batch in flight and the prefetched one
        self._which = 0
        self._prefetched = None        # (tensor, cache index, events) of the last prefetch()
        self.device = torch.device(device)
        self.dtype = dtype
        self.num_class = num_class
        self.layers = []          # (Cin, Cout, stride)
        c = in_channels
        for s in CLS_STRIDES:
            self.layers.append((c, HIDDEN, s))
            c = HIDDEN
        if num_class is not None:
            self.layers.append((HIDDEN * len(CLS_STRIDES), num_class, 1))
        npdt = np.float32 if dtype == torch.float32 else np.float64
        self.filters = [torch.from_numpy(synth.filter_weights(3, 3, 3, ci, co, seed + i, dtype=npdt)).to(self.device)
                        for i, (ci, co, _) in enumerate(self.layers)]
        sizes = [f.numel() for f in self.filters]
        self.fused_grad = torch.zeros(sum(sizes), dtype=dtype, device=self.device)
        self.grad_views = []
        o = 0
        for f, n in zip(self.filters, sizes):
            self.grad_views.append(self.fused_grad[o:o + n].view(f.shape))
            o += n
        self._saved = None
        # description + pointer tables of the stack-level C entry points
        self._desc = _lib.StackDesc()
        self._desc.n_hidden = 4
        self._desc.in_channels = in_channels
        self._desc.hidden = HIDDEN
        self._desc.num_class = int(num_class) if num_class is not None else 0
        self._desc.fz = self._desc.fy = self._desc.fx = 3
        for li, (_, _, s) in enumerate(self.layers):
            for a in range(3):
                self._desc.strides[li][a] = s

    def _fused_flag(self):
        return False if self.fused_launch == "auto" else self.fused_launch

    def fused_status(self):
        """(forward launches, backward launches, error bits) summed over the stack's caches: how many passes ran as ONE
        launch (csrc/conv3p_stack_fused.hpp), and whether any of them reported a barrier time-out (1) or a cloud whose
        tiles did not share an XCC (2).  Synchronises the device."""
        f = b = e = 0
        for c in self._caches:
            if c is not None:
                cf, cb, ce = c.fused_status()
                f, b, e = f + cf, b + cb, e | ce
        return f, b, e

    def _ptr_tables(self):
        """Pointer tables of the stack-level C entry points, rebuilt per call: `filters[i]` may have been rebound
        (an optimizer swap, .to())."""
        nl = len(self.layers)
        for f in self.filters:
            assert f.is_contiguous() and f.device == self.filters[0].device
        return ((ctypes.c_void_p * nl)(*[f.data_ptr() for f in self.filters]),
                (ctypes.c_void_p * nl)(*[g.data_ptr() for g in self.grad_views]))

    def _join_side(self, device):
        """Main stream waits for everything enqueued on the side stream: before a cache that may hold a prefetch in
        flight is overwritten by an inline search, dropped, or handed back to the allocator."""
        if self._side is not None:
            torch.cuda.current_stream(device).wait_stream(self._side)

    # ------------------------------------------------------------------ stack-level C entry points
    def _c_call(self, name, *args):
        lib = _lib.load()
        sfx = "f32" if self.dtype == torch.float32 else "f64"
        rc = getattr(lib, "conv3p_stack_%s_%s" % (name, sfx))(ctypes.byref(self._desc), *args)
        if rc == _lib.ERR_UNSUPPORTED:
            return False
        if rc != _lib.OK:
            raise op.Conv3pRuntimeError("conv3p_stack_%s: %s (status %d)" % (name, _lib.status_string(rc), rc))
        return True

    def _real(self, v):
        return ctypes.c_float(v) if self.dtype == torch.float32 else ctypes.c_double(v)

    def _free_cache_index(self):
        """The cache a new prefetch may use: never the one serving the batch between forward() and backward(); a
        cache that holds a prefetch nobody consumed yet is taken only when nothing else is free (that prefetch
        is then simply lost: its batch will search for itself)."""
        order = (1 - self._which, self._which)
        for idx in order:
            if idx != self._inflight and idx not in self._pending:
                return idx
        for idx in order:
            if idx != self._inflight:
                self._pending.pop(idx, None)
                return idx
        raise op.Conv3pRuntimeError("prefetch(): no neighbour cache is free")   # unreachable with two caches

    def _c_prefetch(self, points):
        for idx, t in self._pending.items():
            if t is points:
                return True
        idx = self._free_cache_index()
        cache = self._cache_slot(idx, points)
        main = torch.cuda.current_stream(points.device)
        if self._side is None:
            self._side = _side_stream(points.device)
        B, N = points.shape[0], points.shape[1]
        with torch.cuda.device(points.device):
            ok = self._c_call("prefetch", points.data_ptr(), self._real(VOXEL), B, N, cache.buf.data_ptr(), cache.nbytes,
                              cache.cfg_ptr(False), self._side.cuda_stream, main.cuda_stream)
        if ok:
            self._pending[idx] = points
        return ok

    def _c_forward(self, points, features):
        B, N = points.shape[0], points.shape[1]
        idx = None
        for i, t in self._pending.items():
            if t is points:
                idx = i
        if idx is None:
            idx = self._which if self._which not in self._pending else 1 - self._which
            if self._pending.pop(idx, None) is not None:
                # a prefetch that was never consumed is overwritten: its searches may still be writing this cache on
                # the side stream, and this batch's search may run on the main stream
                self._join_side(points.device)
        cache = self._cache_slot(idx, points)
        concat = torch.empty((B, N, HIDDEN * 4), dtype=self.dtype, device=points.device)
        head = torch.empty((B, N, self.num_class), dtype=self.dtype, device=points.device) if self.num_class else None
        main = torch.cuda.current_stream(points.device)
        side = None
        if self.overlap_search and idx not in self._pending:
            if self._side is None:
                self._side = _side_stream(points.device)
            side = self._side.cuda_stream
        fptrs, _ = self._ptr_tables()
        with torch.cuda.device(points.device):
            ok = self._c_call("forward", points.data_ptr(), features.data_ptr(), ctypes.cast(fptrs, ctypes.c_void_p),
                              self._real(VOXEL), B, N, concat.data_ptr(), head.data_ptr() if head is not None else None,
                              cache.buf.data_ptr(), cache.nbytes, cache.cfg_ptr(False), main.cuda_stream, side)
        if not ok:
            return None
        self._pending.pop(idx, None)
        self._which = idx
        self._inflight = idx
        self._cache = cache
        acts = [concat[:, :, HIDDEN * i:HIDDEN * (i + 1)] for i in range(4)]
        if head is not None:
            acts.append(head)
        self._saved = (points, features, acts, concat)
        return acts

    def _c_backward(self, upstream):
        points, features, acts, concat = self._saved
        B, N = points.shape[0], points.shape[1]
        lib = _lib.load()
        esz = 4 if self.dtype == torch.float32 else 8
        need = lib.conv3p_stack_scratch_bytes(ctypes.byref(self._desc), esz, B, N)
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(need, dtype=torch.uint8, device=points.device)
        if self.num_class:
            gconcat, ghead = None, upstream[0] if isinstance(upstream, (list, tuple)) else upstream
            ghead = ghead.contiguous()
        else:
            ghead = None
            gconcat = torch.cat(list(upstream), dim=2) if isinstance(upstream, (list, tuple)) else upstream
            gconcat = gconcat.contiguous()
        dx = torch.empty_like(features)
        cache = self._cache
        fptrs, gptrs = self._ptr_tables()
        with torch.cuda.device(points.device):
            ok = self._c_call("backward", points.data_ptr(), features.data_ptr(),
                              ctypes.cast(fptrs, ctypes.c_void_p), self._real(VOXEL), B, N, concat.data_ptr(),
                              acts[4].data_ptr() if self.num_class else None,
                              gconcat.data_ptr() if gconcat is not None else None,
                              ghead.data_ptr() if ghead is not None else None, dx.data_ptr(),
                              ctypes.cast(gptrs, ctypes.c_void_p), self._scratch.data_ptr(), self._scratch.numel(),
                              cache.buf.data_ptr(), cache.nbytes, cache.cfg_ptr(True),
                              torch.cuda.current_stream(points.device).cuda_stream)
        self._inflight = None
        if not ok:
            raise op.Conv3pRuntimeError("conv3p_stack_backward: unsupported after a supported forward")
        return dx, self.fused_grad

    def _cache_slot(self, idx, points):
        B, N = points.shape[0], points.shape[1]
        cmax = max(max(ci, co) for ci, co, _ in self.layers)
        c = self._caches[idx]
        if c is None or not c.fits(B, N, points.dtype, points.device, 27, cmax, cmax):
            if c is not None:
                # the old buffer goes back to the allocator: nothing enqueued on the side stream may still touch it
                self._join_side(points.device)
                self._pending.pop(idx, None)
            c = op.NeighborCache(B, N, points.dtype, points.device, slots=len(self.layers), max_taps=27,
                                 max_cin=cmax, max_cout=cmax, sparse_neighbourhoods=self.sparse_neighbourhoods,
                                 fused_stack=self._fused_flag())
            self._caches[idx] = c
        return c

    def tune(self, points, threshold=32.0):
        """Optional set-up, once per dataset (it synchronises): measure the mean neighbour count of the dilated layers
        on a sample batch and give the stack's caches the matching hint of include/conv3p.h --
        CONV3P_CACHE_SPARSE_NEIGHBOURHOODS when the pair lists are short (ModelNet40-shaped clouds: 7-27 neighbours),
        CONV3P_CACHE_DENSE_NEIGHBOURHOODS when they are long (S3DIS-like rooms: 41-55; measured crossover of the two
        backward kernels between 27 and 41, tools/shape_time.py).  Without tune() the library takes the same decision
        on the device from the lists themselves, at the price of one empty kernel launch per dilated layer and step.
        The hint only selects kernels: results stay within the op's tolerance either way."""
        strides = sorted({s for _, _, s in self.layers if s > 1})
        if not strides or not self.use_cache:
            return None
        sample = points[: min(points.shape[0], 8)].contiguous()
        mean = max(float(op.neighbor_count(sample, (3, 3, 3), (s, s, s), VOXEL).sum(dim=2).float().mean()) for s in strides)
        self.sparse_neighbourhoods = bool(mean <= threshold)
        if self.fused_launch == "auto" and self.sparse_neighbourhoods and self.c_stack:
            self.fused_launch = "forward"   # short lists, even tiles: the forward's hidden layers as one launch
        for c in self._caches:
            if c is not None:
                c.sparse_neighbourhoods = self.sparse_neighbourhoods
                c.fused_stack = self._fused_flag()
        return self.sparse_neighbourhoods

    def prepare(self, B, N):
        """Set-up outside any timed region: allocate both neighbour caches for (B, N) clouds."""
        if self.use_cache:
            for idx in (0, 1):
                cmax = max(max(ci, co) for ci, co, _ in self.layers)
                c = self._caches[idx]
                if c is None or not c.fits(B, N, self.dtype, self.device, 27, cmax, cmax):
                    self._caches[idx] = op.NeighborCache(B, N, self.dtype, self.device, slots=len(self.layers),
                                                         max_taps=27, max_cin=cmax, max_cout=cmax,
                                                         sparse_neighbourhoods=self.sparse_neighbourhoods,
                                                         fused_stack=self._fused_flag())

    def _cache_for(self, points):
        if not self.use_cache:
            return None
        self._cache = self._cache_slot(self._which, points)
        return self._cache

    def prefetch(self, points):
        """Software pipelining across steps: enqueue the geometry (sort + every layer's neighbour search) of the
        NEXT batch on the side stream, typically right after forward() of the current batch, so that it runs under
        the current batch's backward.  Geometry depends on `points` only.  The caller must not modify `points`
        between prefetch() and the forward() that consumes it (that forward trusts the prefetch)."""
        if not self.use_cache:
            return
        if self.c_stack:
            points = points.contiguous()
            if self._c_prefetch(points):
                return
            self.c_stack = False           # shapes the stack-level entry points do not take: op-by-op from now on
        # the cache that holds no pending prefetch: normally the one the current batch is NOT using; when
        # prefetch() is called before forward() of the batch prefetched earlier (i.e. the batch in `_which` is
        # finished: its backward has been enqueued), that finished batch's cache
        pend = self._prefetched
        idx = 1 - self._which
        if pend is not None and pend[1] == idx:
            if pend[0] is points:
                return
            idx = self._which
        cache = self._cache_slot(idx, points)
        self._pending2 = pend                              # keep the earlier prefetch for its forward()
        if not self.batched_prefetch:
            _, events = self._enqueue_searches(points, cache)
            self._prefetched = (points, idx, events)
            return
        # nothing waits for the first layer's lists here, so all layers' searches go out as ONE launch
        main = torch.cuda.current_stream(points.device)
        if self._side is None:
            self._side = _side_stream(points.device)
        self._side.wait_stream(main)                       # `points` is ready on the main stream
        strides = []
        for _, _, s in self.layers:
            if (s, s, s) not in strides:
                strides.append((s, s, s))
        op.cache_prepare_multi(points, (3, 3, 3), strides, VOXEL, cache, stream=self._side)
        ev = torch.cuda.Event()
        ev.record(self._side)
        self._prefetched = (points, idx, [ev] * len(self.layers))

    def _enqueue_searches(self, points, cache):
        """All layers' geometry on the side stream; returns one event per layer."""
        main = torch.cuda.current_stream(points.device)
        if self._side is None:
            self._side = _side_stream(points.device)
        self._side.wait_stream(main)                       # `points` is ready on the main stream
        events = []
        for li, (_, _, s) in enumerate(self.layers):
            op.cache_prepare(points, (3, 3, 3), (s, s, s), VOXEL, cache, points_unchanged=li > 0, stream=self._side)
            ev = torch.cuda.Event()
            ev.record(self._side)
            events.append(ev)
        return main, events

    def forward(self, points, features):
        if self.c_stack:
            acts = self._c_forward(points.contiguous() if not points.is_contiguous() else points,
                                   features.contiguous() if not features.is_contiguous() else features)
            if acts is not None:
                return acts
            self.c_stack = False
        pf = self._prefetched
        p2 = getattr(self, "_pending2", None)
        if p2 is not None and p2[0] is points and self.use_cache:
            pf, keep = p2, self._prefetched               # the older of two outstanding prefetches
            self._pending2 = None
        else:
            keep = None
        if pf is not None and pf[0] is points and self.use_cache:
            # geometry was enqueued by prefetch(): switch to that cache, wait for its per-layer events
            self._which = pf[1]
            cache = self._cache = self._caches[pf[1]]
            main, events = torch.cuda.current_stream(points.device), pf[2]
            self._prefetched = keep
        else:
            cache = self._cache_for(points)
            main, events = (self._enqueue_searches(points, cache) if self.overlap_search else (None, None))
        acts, x = [], features
        for li in range(4):
            _, _, s = self.layers[li]
            if events is not None:
                main.wait_event(events[li])
            hint = li > 0 or events is not None                      # the first call of a step re-validates
            if self.fuse_selu:
                x = op.conv3p_layer(points, x, self.filters[li], (s, s, s), VOXEL, cache, points_unchanged=hint)
            else:
                x = op.selu(op.conv3p(points, x, self.filters[li], (s, s, s), VOXEL, cache=cache, points_unchanged=hint))
            acts.append(x)
        concat = None
        if self.num_class is not None:
            concat = torch.cat(acts, dim=2)
            if events is not None:
                main.wait_event(events[4])
            acts.append(op.selu(op.conv3p(points, concat, self.filters[4], (1, 1, 1), VOXEL, cache=cache,
                                          points_unchanged=True)))
        self._saved = (points, features, acts, concat)
        return acts

    def backward(self, upstream):
        """upstream: classification -> list of 4 tensors dL/d(act_l) (the slices of dL/dconcat);
        segmentation -> [dL/dlogits_act].  Returns (dL/dfeatures, fused weight-gradient buffer)."""
        if self.c_stack:
            return self._c_backward(upstream)
        points, features, acts, concat = self._saved
        cache = self._cache if self.use_cache else None
        if self.num_class is not None:
            g = op.selu_grad(acts[4], upstream[0])
            dconcat, _ = op.conv3p_grad(g, points, concat, self.filters[4], (1, 1, 1), VOXEL,
                                        grad_filter_out=self.grad_views[4], cache=cache, points_unchanged=True)
            ext = [dconcat[:, :, HIDDEN * i:HIDDEN * (i + 1)].contiguous() for i in range(4)]
        elif isinstance(upstream, torch.Tensor):     # dL/dconcat as one (B, N, 36) tensor
            ext = [upstream[:, :, HIDDEN * i:HIDDEN * (i + 1)].contiguous() for i in range(4)]
        else:
            ext = list(upstream)
        if self.fuse_selu:
            # g = dL/d(conv output of layer li); each layer's backward emits the next g directly
            g = op.selu_grad(acts[3], ext[3])
            for li in (3, 2, 1):
                _, _, s = self.layers[li]
                g, _ = op.conv3p_layer_grad(g, points, acts[li - 1], self.filters[li], (s, s, s), VOXEL, cache,
                                            grad_addend=ext[li - 1], grad_filter_out=self.grad_views[li],
                                            points_unchanged=True)
            _, _, s = self.layers[0]
            carry, _ = op.conv3p_grad(g, points, features, self.filters[0], (s, s, s), VOXEL,
                                      grad_filter_out=self.grad_views[0], cache=cache, points_unchanged=True)
            return carry, self.fused_grad
        carry = None
        for li in (3, 2, 1, 0):
            _, _, s = self.layers[li]
            g = op.selu_grad(acts[li], ext[li], carry)
            x_in = acts[li - 1] if li > 0 else features
            carry, _ = op.conv3p_grad(g, points, x_in, self.filters[li], (s, s, s), VOXEL,
                                      grad_filter_out=self.grad_views[li], cache=cache, points_unchanged=True)
        return carry, self.fused_grad


def selu_numpy(x):
    """CPU SELU for the oracle side of stack comparisons (selu.py:22-26)."""
    alpha, scale = 1.6732632423543772848170429916717, 1.0507009873554804934193349852946
    return (scale * np.where(x >= 0, x, alpha * np.expm1(x))).astype(x.dtype)


def selu_grad_numpy(y, dy):
    alpha, scale = 1.6732632423543772848170429916717, 1.0507009873554804934193349852946
    return (dy * np.where(y >= 0, scale, y + scale * alpha)).astype(y.dtype)
