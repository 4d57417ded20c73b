"""Host-side mirror of the reference's operator interface for the conv3p hot path.

The reference loads tf_conv3p.so with tf.load_op_library and calls
    conv3p_module.conv3p(points, input, filter, stride, voxel_size)                  -> output
    conv3p_module.conv3p_grad(grad, points, input, filter, stride, voxel_size)       -> (grad_input, grad_filter)
(/root/reference/pointcnn2_acsd.py:10-13, :30; op schema tf_ops/conv3p/register_op.cpp:44-75), and wires
the gradient in Python returning [None, input_grad, filter_grad, None, None] (pointcnn2_acsd.py:15-31).

This module keeps those names, argument order, shapes and error behaviour, with torch tensors standing in
for TF tensors (torch is only the device-memory / stream plumbing here): every call goes through the C ABI of
include/conv3p.h into the hand-written gfx950 kernels.  There is no CPU implementation in this package; CPU
tensors are rejected.

Shape checks mirror the reference's OP_REQUIRES (tf_conv3p_atrous.cpp:410-443, :549-585) and raise
Conv3pInvalidArgument (the analogue of tensorflow.errors.InvalidArgumentError) with the same messages.
"""
import ctypes

import torch

from . import _lib


class Conv3pInvalidArgument(ValueError):
    """Analogue of tf.errors.InvalidArgumentError raised by the op's OP_REQUIRES checks."""


class Conv3pRuntimeError(RuntimeError):
    pass


_SFX = {torch.float32: ("f32", ctypes.c_float, 4), torch.float64: ("f64", ctypes.c_double, 8)}

# one growing scratch buffer per (device, stream): the caller-owned workspace of the C ABI
_WORKSPACES = {}


def _workspace(device, nbytes):
    stream = torch.cuda.current_stream(device)
    key = (device.index, stream.cuda_stream)
    buf = _WORKSPACES.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _WORKSPACES[key] = buf
    return buf, stream


class NeighborCache:
    """Caller-owned persistent device buffer for the *_cached entry points of include/conv3p.h.

    One cache serves one (B, N, dtype) on one device.  It is zero-filled at creation; validity afterwards
    is decided on the device by content hash, so reusing it with different clouds is always safe."""

    def __init__(self, B, N, dtype, device, slots=5, max_taps=27, pairs_per_point=0, max_cin=36, max_cout=41,
                 sparse_neighbourhoods=None, fused_stack=False, trust_tensor_identity=False):
        lib = _lib.load()
        # trust_tensor_identity (opt-in; what integration/tf_conv3p_shim.cc does with TensorFlow's tensors): the cache keeps
        # a REFERENCE to the points tensor it validated last; a later op call whose `points` is that very storage, at the
        # same address, shape and torch version counter (every in-place write through torch bumps it, and a storage that
        # is referenced cannot be recycled) carries CONV3P_CACHE_POINTS_UNCHANGED by itself -- no hash launch and none of
        # the three launches that would find nothing to do.  Code that writes the buffer behind torch's back (a foreign
        # kernel) must call forget_points() or leave this off.
        self.trust_tensor_identity = trust_tensor_identity
        self._held = None
        # CONV3P_CACHE_FUSED_FORWARD / _BACKWARD for the stack-level entry points (opt-in): True = both, or "forward" / "backward"
        self.fused_stack = fused_stack
        self.cfg = _lib.CacheConfig(slots, max_taps, pairs_per_point, max_cin, max_cout, 0)
        # None: the library decides on the device which backward kernel serves the dilated narrow layers; True / False:
        # CONV3P_CACHE_SPARSE_NEIGHBOURHOODS / CONV3P_CACHE_DENSE_NEIGHBOURHOODS (include/conv3p.h), saving an empty launch
        self.sparse_neighbourhoods = sparse_neighbourhoods
        self.key = (int(B), int(N), dtype, torch.device(device))
        esz = _SFX[dtype][2]
        self.nbytes = lib.conv3p_cache_bytes(esz, B, N, ctypes.byref(self.cfg))
        if self.nbytes == 0:
            raise Conv3pInvalidArgument("bad neighbour-cache configuration")
        self.buf = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)

    def cfg_ptr(self, points_unchanged, deep_orders=False):
        """Address of the config struct for one call; points_unchanged = the caller's promise that `points`
        holds the same bytes as at the previous cached call (CONV3P_CACHE_POINTS_UNCHANGED); deep_orders (prepare
        only) = CONV3P_CACHE_PREPARE_DEEP_ORDERS."""
        self.cfg.flags = (_lib.CACHE_POINTS_UNCHANGED if points_unchanged else 0) | \
                         (0 if self.sparse_neighbourhoods is None else _lib.CACHE_SPARSE_NEIGHBOURHOODS
                          if self.sparse_neighbourhoods else _lib.CACHE_DENSE_NEIGHBOURHOODS) | \
                         (_lib.CACHE_PREPARE_DEEP_ORDERS if deep_orders else 0) | \
                         (_lib.CACHE_FUSED_FORWARD if self.fused_stack in (True, "forward") else 0) | \
                         (_lib.CACHE_FUSED_BACKWARD if self.fused_stack in (True, "backward") else 0)
        return ctypes.addressof(self.cfg)

    def _identity_hint(self, points):
        """True iff `points` is the tensor this cache validated last, unmodified (see trust_tensor_identity); otherwise it
        becomes the held one -- to be called only by an op call that then validates it on the device."""
        if not self.trust_tensor_identity:
            return False
        key = (points.data_ptr(), points.untyped_storage().data_ptr(), int(points._version), tuple(points.shape), points.dtype)
        held = self._held
        if held is not None and held[1] == key and int(held[0]._version) == key[2]:
            return True
        self._held = (points, key)
        return False

    def forget_points(self):
        self._held = None

    def fits(self, B, N, dtype, device, ntap, cin, cout):
        return (self.key == (int(B), int(N), dtype, torch.device(device)) and ntap <= self.cfg.max_taps
                and cin <= self.cfg.max_Cin and cout <= self.cfg.max_Cout)

    def fused_status(self):
        """(forward launches, backward launches, error bits) of the fused stack launches this cache has served
        (conv3p_cache_fused_status; synchronises the device)."""
        f, b, e = ctypes.c_uint(0), ctypes.c_uint(0), ctypes.c_uint(0)
        rc = _lib.load().conv3p_cache_fused_status(self.buf.data_ptr(), ctypes.byref(f), ctypes.byref(b), ctypes.byref(e))
        if rc != _lib.OK:
            raise Conv3pRuntimeError("conv3p_cache_fused_status: status %d" % rc)
        return f.value, b.value, e.value

    def __del__(self):
        try:
            _lib.load().conv3p_cache_forget(self.buf.data_ptr())
        except Exception:
            pass


def _require(cond, msg):
    if not cond:
        raise Conv3pInvalidArgument(msg)


def _stride_list(stride):
    if isinstance(stride, torch.Tensor):
        _require(stride.dim() >= 1 and stride.shape[0] == 3, "Conv3p expects stride tensor to have size 3.")
        stride = stride.detach().cpu().tolist()   # host-memory input in the TF op (read on the host, .cpp:438-440)
    stride = [int(s) for s in stride]
    _require(len(stride) == 3, "Conv3p expects stride tensor to have size 3.")
    return (ctypes.c_int32 * 3)(*stride)


def _voxel_value(voxel_size):
    if isinstance(voxel_size, torch.Tensor):
        _require(voxel_size.dim() >= 1 and voxel_size.shape[0] == 1, "Conv3p expects voxel tensor to have dimension 1.")
        return float(voxel_size.detach().cpu().reshape(-1)[0])
    if isinstance(voxel_size, (list, tuple)):
        _require(len(voxel_size) == 1, "Conv3p expects voxel tensor to have dimension 1.")
        return float(voxel_size[0])
    return float(voxel_size)


def _check_device(*tensors):
    dev = tensors[0].device
    if dev.type != "cuda":
        raise Conv3pRuntimeError("conv3p: tensors must live on a HIP device (no CPU path in pointwise_amd)")
    for t in tensors:
        if t.device != dev:
            raise Conv3pRuntimeError("conv3p: all tensors must be on the same device")
    return dev


def _common_checks(points, input, filter):
    _require(points.dim() == 3, "Conv3p expects (batch_size, num_points, 3) points shape")
    _require(points.shape[2] == 3, "Conv3p expects (batch_size, num_points, 3) points shape")
    _require(input.dim() == 3 and input.shape[0] == points.shape[0],
             "Conv3p expects points and input tensor to have the same batch size")
    _require(input.shape[1] == points.shape[1],
             "Conv3p expects points and input tensor to have the same number of points")
    _require(filter.dim() == 5, "Conv3p expects [filter_z, filter_y, filter_x, in_channels, out_channels] filter")
    _require(filter.shape[3] == input.shape[2], "Conv3p expects filter channels to be matched with input channels")
    dt = points.dtype
    if dt not in _SFX:
        raise Conv3pInvalidArgument("Conv3p: T must be float32 or float64")   # Attr T: {float, double}
    _require(input.dtype == dt and filter.dtype == dt, "Conv3p: points, input and filter must share dtype T")


def _call(fn, *args):
    rc = fn(*args)
    if rc == _lib.OK:
        return
    msg = "conv3p: %s (status %d)" % (_lib.status_string(rc), rc)
    if rc == _lib.ERR_INVALID_ARGUMENT:
        raise Conv3pInvalidArgument(msg)
    raise Conv3pRuntimeError(msg)


def conv3p(points, input, filter, stride, voxel_size, cache=None, points_unchanged=False, _fused_selu=False):
    """Conv3p forward.  points (B,N,3), input (B,N,Cin), filter (fz,fy,fx,Cin,Cout), stride [sx,sy,sz]
    (int32[3]), voxel_size T[1] -> output (B,N,Cout).  Mirrors conv3p_module.conv3p
    (/root/reference/pointcnn2_acsd.py:12-13).  cache (optional, not in the reference): a NeighborCache."""
    lib = _lib.load()
    _common_checks(points, input, filter)
    s3 = _stride_list(stride)
    vox = _voxel_value(voxel_size)
    dev = _check_device(points, input, filter)
    sfx, creal, esz = _SFX[points.dtype]
    B, N, _ = points.shape
    fz, fy, fx, Cin, Cout = filter.shape
    points, input, filter = points.contiguous(), input.contiguous(), filter.contiguous()
    out = torch.empty((B, N, Cout), dtype=points.dtype, device=dev)
    with torch.cuda.device(dev):
        if cache is not None and cache.fits(B, N, points.dtype, dev, fz * fy * fx, Cin, Cout):
            stream = torch.cuda.current_stream(dev)
            _call(getattr(lib, ("conv3p_layer_forward_cached_" if _fused_selu else "conv3p_forward_cached_") + sfx),
                  points.data_ptr(), input.data_ptr(),
                  filter.data_ptr(), ctypes.cast(s3, ctypes.c_void_p), creal(vox), B, N, Cin, Cout, fz, fy, fx,
                  out.data_ptr(), cache.buf.data_ptr(), cache.nbytes,
                  cache.cfg_ptr(points_unchanged or cache._identity_hint(points)), stream.cuda_stream)
        else:
            if _fused_selu:
                raise Conv3pInvalidArgument("conv3p_layer needs a NeighborCache that fits these clouds")
            need = lib.conv3p_workspace_bytes(_lib.PASS_FORWARD, esz, B, N, Cin, Cout, fz, fy, fx)
            ws, stream = _workspace(dev, need)
            _call(getattr(lib, "conv3p_forward_" + sfx), points.data_ptr(), input.data_ptr(), filter.data_ptr(),
                  ctypes.cast(s3, ctypes.c_void_p), creal(vox), B, N, Cin, Cout, fz, fy, fx, out.data_ptr(),
                  ws.data_ptr(), ws.numel(), stream.cuda_stream)
    return out


def conv3p_grad(grad_from_next, points, input, filter, stride, voxel_size, grad_filter_out=None, cache=None,
                points_unchanged=False, _fused_selu=False, _grad_addend=None):
    """Conv3pGrad -> (grad_input, grad_filter).  Mirrors conv3p_module.conv3p_grad
    (/root/reference/pointcnn2_acsd.py:30; schema register_op.cpp:63-75).
    grad_filter_out (optional, not in the reference): a contiguous tensor shaped like filter to write
    grad_filter into, e.g. a view of the fused buffer that is all-reduced across GPUs."""
    lib = _lib.load()
    _common_checks(points, input, filter)
    s3 = _stride_list(stride)
    vox = _voxel_value(voxel_size)
    B, N, _ = points.shape
    fz, fy, fx, Cin, Cout = filter.shape
    _require(grad_from_next.dim() == 3 and grad_from_next.shape[0] == B, "backprop grad tensor has wrong size for dim 0")
    _require(grad_from_next.shape[1] == N, "backprop grad tensor has wrong size for dim 1")
    _require(grad_from_next.shape[2] == Cout, "backprop grad tensor has wrong size for dim 2")
    _require(grad_from_next.dtype == points.dtype, "Conv3pGrad: grad_from_next must have dtype T")
    dev = _check_device(grad_from_next, points, input, filter)
    sfx, creal, esz = _SFX[points.dtype]
    grad_from_next = grad_from_next.contiguous()
    points, input, filter = points.contiguous(), input.contiguous(), filter.contiguous()
    dx = torch.empty_like(input)
    if grad_filter_out is None:
        dw = torch.empty_like(filter)
    else:
        dw = grad_filter_out
        if dw.shape != filter.shape or dw.dtype != filter.dtype or not dw.is_contiguous() or dw.device != dev:
            raise Conv3pInvalidArgument("grad_filter_out must be a contiguous tensor like filter")
    with torch.cuda.device(dev):
        if cache is not None and cache.fits(B, N, points.dtype, dev, fz * fy * fx, Cin, Cout):
            stream = torch.cuda.current_stream(dev)
            if _fused_selu:
                add = None
                if _grad_addend is not None:
                    add = _grad_addend
                    if add.shape != input.shape or add.dtype != input.dtype or add.device != dev:
                        raise Conv3pInvalidArgument("grad_addend must be a tensor like input")
                    add = add.contiguous()
                _call(getattr(lib, "conv3p_layer_backward_cached_" + sfx), grad_from_next.data_ptr(),
                      points.data_ptr(), input.data_ptr(), filter.data_ptr(), ctypes.cast(s3, ctypes.c_void_p),
                      creal(vox), B, N, Cin, Cout, fz, fy, fx, add.data_ptr() if add is not None else None,
                      dx.data_ptr(), dw.data_ptr(), cache.buf.data_ptr(), cache.nbytes,
                      cache.cfg_ptr(points_unchanged or cache._identity_hint(points)), stream.cuda_stream)
            else:
                _call(getattr(lib, "conv3p_backward_cached_" + sfx), grad_from_next.data_ptr(), points.data_ptr(),
                      input.data_ptr(), filter.data_ptr(), ctypes.cast(s3, ctypes.c_void_p), creal(vox), B, N, Cin,
                      Cout, fz, fy, fx, dx.data_ptr(), dw.data_ptr(), cache.buf.data_ptr(), cache.nbytes,
                      cache.cfg_ptr(points_unchanged or cache._identity_hint(points)), stream.cuda_stream)
        else:
            if _fused_selu:
                raise Conv3pInvalidArgument("conv3p_layer_grad needs a NeighborCache that fits these clouds")
            need = lib.conv3p_workspace_bytes(_lib.PASS_BACKWARD, esz, B, N, Cin, Cout, fz, fy, fx)
            ws, stream = _workspace(dev, need)
            _call(getattr(lib, "conv3p_backward_" + sfx), grad_from_next.data_ptr(), points.data_ptr(),
                  input.data_ptr(), filter.data_ptr(), ctypes.cast(s3, ctypes.c_void_p), creal(vox), B, N, Cin,
                  Cout, fz, fy, fx, dx.data_ptr(), dw.data_ptr(), ws.data_ptr(), ws.numel(), stream.cuda_stream)
    return dx, dw


def conv3p_layer(points, input, filter, stride, voxel_size, cache, points_unchanged=False):
    """selu(conv3p(points, input, filter, stride, voxel_size)): the models' layer
    (/root/reference/pointcnn2_acsd.py:48-49), SELU fused into the op (conv3p_layer_forward_cached_*)."""
    return conv3p(points, input, filter, stride, voxel_size, cache=cache, points_unchanged=points_unchanged,
                  _fused_selu=True)


def conv3p_layer_grad(grad_from_next, points, input, filter, stride, voxel_size, cache, grad_addend=None,
                      grad_filter_out=None, points_unchanged=False):
    """Backward of a layer whose `input` is a SELU output (conv3p_layer_backward_cached_*):
    returns ((dX + grad_addend) * selu'(input), grad_filter) where (dX, grad_filter) = conv3p_grad(...), i.e.
    the gradient w.r.t. the argument of the SELU that produced `input`; grad_addend is the gradient `input`
    receives from its other consumer (the concat of pointcnn2_acsd.py:66)."""
    return conv3p_grad(grad_from_next, points, input, filter, stride, voxel_size, grad_filter_out=grad_filter_out,
                       cache=cache, points_unchanged=points_unchanged, _fused_selu=True, _grad_addend=grad_addend)


def cache_prepare(points, filter_zyx, stride, voxel_size, cache, points_unchanged=False, stream=None,
                  deep_orders=False):
    """Build / re-validate the geometry of one stencil in `cache` (conv3p_cache_prepare_*), on `stream`
    (default: the current stream).  deep_orders: also the record orders of the matrix-core path (wide fp32 layers), so
    that the layer's forward / backward on these points (points_unchanged=True) need not build them."""
    lib = _lib.load()
    _require(points.dim() == 3 and points.shape[2] == 3, "Conv3p expects (batch_size, num_points, 3) points shape")
    dev = _check_device(points)
    sfx, creal, esz = _SFX[points.dtype]
    s3 = _stride_list(stride)
    vox = _voxel_value(voxel_size)
    B, N, _ = points.shape
    fz, fy, fx = [int(v) for v in filter_zyx]
    if not cache.fits(B, N, points.dtype, dev, fz * fy * fx, 0, 0):
        raise Conv3pInvalidArgument("neighbour cache does not fit these clouds")
    points = points.contiguous()
    with torch.cuda.device(dev):
        st = stream if stream is not None else torch.cuda.current_stream(dev)
        _call(getattr(lib, "conv3p_cache_prepare_" + sfx), points.data_ptr(), ctypes.cast(s3, ctypes.c_void_p),
              creal(vox), B, N, fz, fy, fx, cache.buf.data_ptr(), cache.nbytes,
              cache.cfg_ptr(points_unchanged, deep_orders=deep_orders and points.dtype == torch.float32), st.cuda_stream)


def cache_prepare_multi(points, filter_zyx, strides, voxel_size, cache, points_unchanged=False, stream=None):
    """cache_prepare for several strides at once (conv3p_cache_prepare_multi_*): one sort, one search launch and
    one normaliser launch for all of them."""
    lib = _lib.load()
    _require(points.dim() == 3 and points.shape[2] == 3, "Conv3p expects (batch_size, num_points, 3) points shape")
    dev = _check_device(points)
    sfx, creal, esz = _SFX[points.dtype]
    vox = _voxel_value(voxel_size)
    B, N, _ = points.shape
    fz, fy, fx = [int(v) for v in filter_zyx]
    flat = []
    for st in strides:
        flat.extend(list(_stride_list(st)))
    K = len(flat) // 3
    arr = (ctypes.c_int32 * len(flat))(*flat)
    if not cache.fits(B, N, points.dtype, dev, fz * fy * fx, 0, 0):
        raise Conv3pInvalidArgument("neighbour cache does not fit these clouds")
    points = points.contiguous()
    with torch.cuda.device(dev):
        st_ = stream if stream is not None else torch.cuda.current_stream(dev)
        _call(getattr(lib, "conv3p_cache_prepare_multi_" + sfx), points.data_ptr(), ctypes.cast(arr, ctypes.c_void_p),
              K, creal(vox), B, N, fz, fy, fx, cache.buf.data_ptr(), cache.nbytes, cache.cfg_ptr(points_unchanged),
              st_.cuda_stream)


def neighbor_count(points, filter_zyx, stride, voxel_size):
    """int32 (B, N, fz*fy*fx) per-tap neighbour populations (the op's normaliser), for exact parity checks."""
    lib = _lib.load()
    _require(points.dim() == 3 and points.shape[2] == 3, "Conv3p expects (batch_size, num_points, 3) points shape")
    dev = _check_device(points)
    if points.dtype not in _SFX:
        raise Conv3pInvalidArgument("Conv3p: T must be float32 or float64")
    sfx, creal, esz = _SFX[points.dtype]
    s3 = _stride_list(stride)
    vox = _voxel_value(voxel_size)
    B, N, _ = points.shape
    fz, fy, fx = [int(v) for v in filter_zyx]
    points = points.contiguous()
    cnt = torch.zeros((B, N, fz * fy * fx), dtype=torch.int32, device=dev)
    need = lib.conv3p_workspace_bytes(_lib.PASS_NEIGHBOR_COUNT, esz, B, N, 0, 0, fz, fy, fx)
    with torch.cuda.device(dev):
        ws, stream = _workspace(dev, need)
        _call(getattr(lib, "conv3p_neighbor_count_" + sfx), points.data_ptr(), ctypes.cast(s3, ctypes.c_void_p),
              creal(vox), B, N, fz, fy, fx, cnt.data_ptr(), ws.data_ptr(), ws.numel(), stream.cuda_stream)
    return cnt


def selu(x):
    """SELU as the reference applies it after each conv3p (/root/reference/selu.py:22-26)."""
    lib = _lib.load()
    dev = _check_device(x)
    sfx = _SFX[x.dtype][0]
    x = x.contiguous()
    y = torch.empty_like(x)
    with torch.cuda.device(dev):
        _call(getattr(lib, "conv3p_selu_" + sfx), x.data_ptr(), y.data_ptr(), x.numel(),
              torch.cuda.current_stream(dev).cuda_stream)
    return y


def selu_grad(y, dy, dy_b=None):
    """dL/dx of SELU given its output y and dL/dy (= dy, or dy + dy_b when the activation has two consumers)."""
    lib = _lib.load()
    dev = _check_device(y, dy)
    sfx = _SFX[y.dtype][0]
    y, dy = y.contiguous(), dy.contiguous()
    dx = torch.empty_like(y)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream(dev).cuda_stream
        if dy_b is None:
            _call(getattr(lib, "conv3p_selu_grad_" + sfx), y.data_ptr(), dy.data_ptr(), dx.data_ptr(), y.numel(),
                  stream)
        else:
            dy_b = dy_b.contiguous()
            _call(getattr(lib, "conv3p_selu_grad_add_" + sfx), y.data_ptr(), dy.data_ptr(), dy_b.data_ptr(),
                  dx.data_ptr(), y.numel(), stream)
    return dx


class Conv3pFunction(torch.autograd.Function):
    """Autograd wiring identical to the reference's @tf.RegisterGradient('Conv3p')
    (/root/reference/pointcnn2_acsd.py:15-31): gradients [None, input_grad, filter_grad, None, None]."""

    @staticmethod
    def forward(ctx, points, input, filter, stride, voxel_size):
        ctx.save_for_backward(points, input, filter)
        ctx.stride = stride
        ctx.voxel_size = voxel_size
        return conv3p(points, input, filter, stride, voxel_size)

    @staticmethod
    def backward(ctx, grad_from_next_layer):
        points, input, filter = ctx.saved_tensors
        input_grad, filter_grad = conv3p_grad(grad_from_next_layer, points, input, filter, ctx.stride,
                                              ctx.voxel_size)
        return None, input_grad, filter_grad, None, None


def conv3p_autograd(points, input, filter, stride, voxel_size):
    return Conv3pFunction.apply(points, input, filter, stride, voxel_size)
