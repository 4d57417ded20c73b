"""float64 restatement of the classification model's dense head -- TEST INFRASTRUCTURE ONLY.

/root/reference/pointcnn2_acsd.py:68-90 with /root/reference/selu.py:22-26 (selu) and :35-70 (dropout_selu).
tf.contrib.layers.fully_connected is a third-party (TensorFlow 1.x contrib) layer that is not in the reference tree
or in this image; its published definition is outputs = activation_fn(inputs . weights + biases), restated here.
PARITY STATUS: unpinned (no TensorFlow to run, the reference ships no vectors for the head); the HIP kernels are
checked against this float64 restatement and by finite differences.
"""
import numpy as np

ALPHA = 1.6732632423543772848170429916717
SCALE = 1.0507009873554804934193349852946


def selu(x):
    return SCALE * np.where(x >= 0.0, x, ALPHA * np.expm1(x))                      # selu.py:22-26


def selu_slope_from_output(y):
    return np.where(y >= 0.0, SCALE, y + SCALE * ALPHA)


def fully_connected(x, W, b, act=True):
    z = x.astype(np.float64) @ W.astype(np.float64) + (0.0 if b is None else b.astype(np.float64))
    return selu(z) if act else z


def fully_connected_grad(x, W, y, dy, act=True):
    dz = dy.astype(np.float64) * (selu_slope_from_output(y.astype(np.float64)) if act else 1.0)
    return dz @ W.astype(np.float64).T, x.astype(np.float64).T @ dz, dz.sum(axis=0)


def dropout_selu(x, rate, keep_mask, alpha=-1.7580993408473766, fixed_mean=0.0, fixed_var=1.0):
    keep = 1.0 - rate                                                             # selu.py:40
    ret = x * keep_mask + alpha * (1 - keep_mask)                                 # :56
    a = np.sqrt(fixed_var / (keep * ((1 - keep) * (alpha - fixed_mean) ** 2 + fixed_var)))   # :59
    b = fixed_mean - a * (keep * fixed_mean + (1 - keep) * alpha)                 # :61
    return a * ret + b, a                                                         # :62


def head_forward_backward(feat, W1, b1, W2, b2, labels, rate, keep_mask):
    B = feat.shape[0]
    view = feat.reshape(B, -1).astype(np.float64)
    fc1 = fully_connected(view, W1, b1)
    drop, a = dropout_selu(fc1, rate, keep_mask)
    fc2 = fully_connected(drop, W2, b2)
    z = fc2 - fc2.max(axis=1, keepdims=True)
    logp = z - np.log(np.exp(z).sum(axis=1, keepdims=True))
    loss = -logp[np.arange(B), labels].mean()                                     # pointcnn2_acsd.py:87-88
    dlogits = np.exp(logp)
    dlogits[np.arange(B), labels] -= 1.0
    dlogits /= B
    ddrop, dW2, db2 = fully_connected_grad(drop, W2, fc2, dlogits)
    dfc1 = ddrop * (a * keep_mask)
    dview, dW1, db1 = fully_connected_grad(view, W1, fc1, dfc1)
    return dict(fc1=fc1, logits=fc2, loss=loss, dlogits=dlogits, dfeat=dview.reshape(feat.shape), dW1=dW1, db1=db1,
                dW2=dW2, db2=db2)
