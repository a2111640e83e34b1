"""Keras-semantics layer primitives on torch CPU tensors (NHWC), the oracle for the HIP kernels.

Semantics follow SURVEY.md Appendix B (TensorFlow is unavailable, so these are restatements
of the documented tf.keras behaviour used by the reference's layer calls, e.g.
yolov3/models/backbone.py:27-71, yolov4/models/backbone.py:22-185,
yolov2/models/backbone.py:11-73, yolov1_5/models/backbone.py:9-48).
"""
import math

import torch
import torch.nn.functional as F

BN_EPS = 1e-3
BN_MOMENTUM = 0.99


def same_pad(size, k, s):
    """Keras 'same': out = ceil(size/s); total pad split with the smaller half first."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2, total - total // 2


def conv2d(x, w_hwio, bias=None, stride=1, padding="same"):
    """x: [N,H,W,Cin]; w_hwio: Keras kernel [kh,kw,Cin,Cout]; padding 'same'|'valid'|'darknet_s2'.

    'darknet_s2' = ZeroPadding2D(((1,0),(1,0))) followed by a 'valid' conv
    (yolov3/models/backbone.py:31-34,61)."""
    kh, kw = w_hwio.shape[0], w_hwio.shape[1]
    n, h, wd, c = x.shape
    if padding == "same":
        _, pt, pb = same_pad(h, kh, stride)
        _, pl, pr = same_pad(wd, kw, stride)
    elif padding == "valid":
        pt = pb = pl = pr = 0
    elif padding == "darknet_s2":
        pt, pb, pl, pr = 1, 0, 1, 0
    else:
        raise ValueError(padding)
    xc = x.permute(0, 3, 1, 2)
    xc = F.pad(xc, (pl, pr, pt, pb))
    wc = w_hwio.permute(3, 2, 0, 1)
    y = F.conv2d(xc, wc, bias, stride=stride)
    return y.permute(0, 2, 3, 1).contiguous()


def batchnorm_train(x, gamma, beta, eps=BN_EPS):
    """Training-mode BatchNormalization over (N,H,W): biased batch variance.
    Returns (y, mean, var)."""
    mean = x.mean(dim=(0, 1, 2))
    var = x.var(dim=(0, 1, 2), unbiased=False)
    y = (x - mean) / torch.sqrt(var + eps) * gamma + beta
    return y, mean, var


def batchnorm_infer(x, gamma, beta, moving_mean, moving_var, eps=BN_EPS):
    return (x - moving_mean) / torch.sqrt(moving_var + eps) * gamma + beta


def moving_update(moving, batch, momentum=BN_MOMENTUM):
    return momentum * moving + (1.0 - momentum) * batch


def leaky(x, alpha=0.1):
    return torch.where(x > 0, x, alpha * x)


def leaky_masked(x, positive_mask, alpha=0.1):
    """LeakyReLU with the branch chosen by `positive_mask` instead of by sign(x).

    Test device for end-to-end gradient parity: an fp32 execution and this fp64 oracle can disagree
    on sign(x) only where |x| is within fp32 rounding of 0 (the caller asserts that); forcing the
    device's branch pattern changes the forward value by <= 0.9*|x| ~ 1e-6 there but removes the
    O(1) gradient discontinuity that any two executions of a ReLU network are subject to."""
    return torch.where(positive_mask, x, alpha * x)


def softplus(x):
    return torch.clamp(x, min=0) + torch.log1p(torch.exp(-x.abs()))


def mish(x):
    """yolov4/models/backbone.py:22-37: x * tanh(softplus(x))."""
    return x * torch.tanh(softplus(x))


def upsample2x(x):
    """UpSampling2D(2), nearest."""
    return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)


def maxpool(x, k, stride, padding="valid"):
    """MaxPooling2D; 'same' pads with -inf (smaller half first), 'valid' floors."""
    n, h, w, c = x.shape
    xc = x.permute(0, 3, 1, 2)
    if padding == "same":
        _, pt, pb = same_pad(h, k, stride)
        _, pl, pr = same_pad(w, k, stride)
        xc = F.pad(xc, (pl, pr, pt, pb), value=-math.inf)
    y = F.max_pool2d(xc, k, stride)
    return y.permute(0, 2, 3, 1).contiguous()


def space_to_depth2(x):
    """tf.nn.space_to_depth(x, 2): out[..., (dy*2+dx)*C + c] = x[:, 2h+dy, 2w+dx, c]."""
    n, h, w, c = x.shape
    x = x.reshape(n, h // 2, 2, w // 2, 2, c)
    x = x.permute(0, 1, 3, 2, 4, 5)
    return x.reshape(n, h // 2, w // 2, 4 * c).contiguous()
