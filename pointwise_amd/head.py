"""The dense head of the reference's classification model, on the hand-written FC kernels.

/root/reference/pointcnn2_acsd.py:68-77
    feat = concat(relu1..relu4)                 (B, N, 36)       -- the conv3p stack's output (stack.Conv3pStack)
    view = reshape(feat, [-1, n * 36])          (B, N*36)
    fc1  = fully_connected(view, 512, selu)     W1: (N*36, 512)  -- 151 MB for N = 2048: the model's big object
    drop = dropout_selu(fc1, rate 0.5, training)                 -- selu.py:35-70
    fc2  = fully_connected(drop, num_class, selu)
loss: mean sparse softmax cross-entropy (pointcnn2_acsd.py:79-90).

tf.contrib.layers.fully_connected(x, n, activation_fn) is activation(x . W + b).  Both layers run through
conv3p_fc_forward_f32 / conv3p_fc_backward_f32 (include/conv3p.h): W streamed once per pass, exact fp32 products on
v_mfma_f32_32x32x2_f32, bitwise reproducible.  dropout_selu and the loss are a handful of elementwise torch ops on
(B, 512) / (B, num_class) tensors (device plumbing, no arithmetic worth a kernel).

Data parallelism (one process per GPU): the gradient of W1 is 151 MB per rank -- four orders of magnitude more than
the conv3p filters' 29 KB.  `sharded_gradient_step` reduce-scatters it (every rank receives the SUM of one 1/world
slice: half the traffic of an all-reduce), lets the caller update only that slice (optimizer state is sharded the
same way) and all-gathers the updated weights; launched right after the head's backward on its own stream it runs
under the whole conv3p backward.
"""
import ctypes

import numpy as np
import torch

from . import _lib
from .conv3p_op import Conv3pInvalidArgument, _call, _check_device

SELU_ALPHA = 1.6732632423543772848170429916717
SELU_SCALE = 1.0507009873554804934193349852946

_WS = {}


def _workspace(dev, nbytes):
    key = (dev.index, torch.cuda.current_stream(dev).cuda_stream)
    buf = _WS.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=dev)
        _WS[key] = buf
    return buf


def _check_fc(x, W, b):
    if x.dim() != 2 or W.dim() != 2 or x.shape[1] != W.shape[0]:
        raise Conv3pInvalidArgument("fully_connected: x (M, K) and W (K, N) expected")
    if b is not None and tuple(b.shape) != (W.shape[1],):
        raise Conv3pInvalidArgument("fully_connected: bias must have N entries")
    for t in (x, W) + ((b,) if b is not None else ()):
        if t.dtype != torch.float32:
            raise Conv3pInvalidArgument("fully_connected: float32 only")


def fully_connected(x, W, b=None, selu=True):
    """activation(x . W + b), activation = SELU or identity (tf.contrib.layers.fully_connected)."""
    lib = _lib.load()
    _check_fc(x, W, b)
    dev = _check_device(x, W)
    M, K = x.shape
    N = W.shape[1]
    x, W = x.contiguous(), W.contiguous()
    y = torch.empty((M, N), dtype=torch.float32, device=dev)
    need = lib.conv3p_fc_workspace_bytes(M, K, N)
    with torch.cuda.device(dev):
        ws = _workspace(dev, need)
        _call(lib.conv3p_fc_forward_f32, x.data_ptr(), W.data_ptr(), b.data_ptr() if b is not None else None, M, K, N,
              1 if selu else 0, y.data_ptr(), ws.data_ptr(), ws.numel(), torch.cuda.current_stream(dev).cuda_stream)
    return y


def fully_connected_grad(x, W, y, dy, selu=True, need_dx=True, dW_out=None, db_out=None):
    """Gradients of fully_connected given its OUTPUT y and dL/dy -> (dx or None, dW, db)."""
    lib = _lib.load()
    _check_fc(x, W, None)
    dev = _check_device(x, W, y, dy)
    M, K = x.shape
    N = W.shape[1]
    x, W, y, dy = x.contiguous(), W.contiguous(), y.contiguous(), dy.contiguous()
    dx = torch.empty_like(x) if need_dx else None
    dW = dW_out if dW_out is not None else torch.empty_like(W)
    db = db_out if db_out is not None else torch.empty((N,), dtype=torch.float32, device=dev)
    if not dW.is_contiguous() or tuple(dW.shape) != (K, N):
        raise Conv3pInvalidArgument("dW_out must be a contiguous (K, N) tensor")
    need = lib.conv3p_fc_workspace_bytes(M, K, N)
    with torch.cuda.device(dev):
        ws = _workspace(dev, need)
        _call(lib.conv3p_fc_backward_f32, x.data_ptr(), W.data_ptr(), y.data_ptr(), dy.data_ptr(), M, K, N,
              1 if selu else 0, dx.data_ptr() if dx is not None else None, dW.data_ptr(), db.data_ptr(),
              ws.data_ptr(), ws.numel(), torch.cuda.current_stream(dev).cuda_stream)
    return dx, dW, db


def dropout_selu_constants(rate, alpha=-1.7580993408473766, fixed_mean=0.0, fixed_var=1.0):
    """a, b of selu.py:59-61 for keep_prob = 1 - rate."""
    keep = 1.0 - rate
    a = np.sqrt(fixed_var / (keep * ((1 - keep) * (alpha - fixed_mean) ** 2 + fixed_var)))
    b = fixed_mean - a * (keep * fixed_mean + (1 - keep) * alpha)
    return float(a), float(b), float(alpha)


def dropout_selu(x, rate, training, keep_mask=None):
    """selu.py:35-70: ret = a * (x * keep + alpha' * (1 - keep)) + b in training, identity otherwise.
    keep_mask (0/1 tensor like x): the Bernoulli(keep_prob) draw; None draws floor(keep_prob + U[0,1))."""
    if not training or rate == 0.0:
        return x, None
    a, b, alpha = dropout_selu_constants(rate)
    if keep_mask is None:
        keep_mask = torch.floor((1.0 - rate) + torch.rand_like(x))            # selu.py:53-55
    return a * (x * keep_mask + alpha * (1 - keep_mask)) + b, keep_mask        # :56, :62


class ClassificationHead:
    """fc1 (N*36 -> 512, selu) -> dropout_selu -> fc2 (512 -> num_class, selu); parameters live on `device`."""

    def __init__(self, n_points, num_class=40, feat_channels=36, hidden=512, device="cuda:0", seed=0, rate=0.5):
        g = torch.Generator().manual_seed(seed)
        K = n_points * feat_channels
        # variance-scaling (FAN_IN) initialisation as recommended next to selu (selu.py:29-31); zero biases
        self.W1 = (torch.randn((K, hidden), generator=g) * (1.0 / np.sqrt(K))).to(device)
        self.b1 = torch.zeros(hidden, device=device)
        self.W2 = (torch.randn((hidden, num_class), generator=g) * (1.0 / np.sqrt(hidden))).to(device)
        self.b2 = torch.zeros(num_class, device=device)
        self.rate = rate
        self.dW1 = torch.empty_like(self.W1)           # 151 MB for N = 2048: allocated once
        self.db1 = torch.empty_like(self.b1)
        self.dW2 = torch.empty_like(self.W2)
        self.db2 = torch.empty_like(self.b2)
        self._saved = None

    def forward(self, feat, training=True, keep_mask=None):
        B = feat.shape[0]
        view = feat.reshape(B, -1)                                              # pointcnn2_acsd.py:70
        fc1 = fully_connected(view, self.W1, self.b1, selu=True)                # :71
        drop, mask = dropout_selu(fc1, self.rate, training, keep_mask)          # :73
        fc2 = fully_connected(drop, self.W2, self.b2, selu=True)                # :75
        self._saved = (view, fc1, drop, mask, fc2, feat.shape)
        return fc2

    def loss(self, logits, labels, global_batch=None):
        """mean sparse softmax cross-entropy (pointcnn2_acsd.py:79-90) and its gradient w.r.t. the logits.

        Data parallel: pass global_batch = the batch over ALL ranks.  distributed.py reduces gradients with SUM, so the
        gradient of the reference's mean over the global batch is each rank's dlogits / global_batch (dividing by
        the local batch would scale every gradient by the world size).  The returned loss is the local shard's mean."""
        logp = torch.log_softmax(logits, dim=1)
        idx = labels.long().unsqueeze(1)
        e = -(logp.gather(1, idx)).mean()
        dlogits = torch.softmax(logits, dim=1)
        dlogits.scatter_add_(1, idx, -torch.ones_like(idx, dtype=dlogits.dtype))
        return e, dlogits / float(global_batch if global_batch is not None else logits.shape[0])

    def backward(self, dlogits):
        """-> dL/dfeat (B, N, 36); parameter gradients in self.dW1, db1, dW2, db2."""
        view, fc1, drop, mask, fc2, shape = self._saved
        ddrop, _, _ = fully_connected_grad(drop, self.W2, fc2, dlogits, selu=True, dW_out=self.dW2, db_out=self.db2)
        if mask is not None:
            a, _, _ = dropout_selu_constants(self.rate)
            dfc1 = ddrop * (a * mask)
        else:
            dfc1 = ddrop
        dview, _, _ = fully_connected_grad(view, self.W1, fc1, dfc1, selu=True, dW_out=self.dW1, db_out=self.db1)
        return dview.reshape(shape)

    def parameters(self):
        return [self.W1, self.b1, self.W2, self.b2]

    def gradients(self):
        return [self.dW1, self.db1, self.dW2, self.db2]
