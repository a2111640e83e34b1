"""Known-answer cases derived BY HAND from the reference's source lines (no oracle, no device involved): every
expectation below is plain Python arithmetic on a handful of numbers, written next to the reference line it follows.
Both the CPU oracle (tests/test_kat_oracle_cpu.py) and the HIP kernels through the C-ABI (tests/test_gpu_kats.py)
must reproduce them, which pins the tf.keras half of the oracle as far as the source allows without TensorFlow.

All label / prediction tensors are NHWC float32 exactly as the reference's closures receive them.
"""
import math

import numpy as np

EPS = 1e-07   # EPSILON of every reference loss / metric file


def _iou_xywh(t, p):
    """cal_iou for one box pair whose centres are already divided by the grid (yolov3/losses/loss.py:9-37)"""
    tx0, tx1, ty0, ty1 = t[0] - t[2] / 2, t[0] + t[2] / 2, t[1] - t[3] / 2, t[1] + t[3] / 2
    px0, px1, py0, py1 = p[0] - p[2] / 2, p[0] + p[2] / 2, p[1] - p[3] / 2, p[1] + p[3] / 2
    iw = max(min(tx1, px1) - max(tx0, px0), 0.0)
    ih = max(min(ty1, py1) - max(ty0, py0), 0.0)
    inter = iw * ih
    return inter / (t[2] * t[3] + p[2] * p[3] - inter + EPS)


# ------------------------------------------------------------------------------------------------------------------
# YOLOv2 loss, yolov2/losses/loss.py:75-133
# ------------------------------------------------------------------------------------------------------------------
V2_ANCHORS5 = [(0.04405615, 0.05210654), (0.14418923, 0.15865615), (0.25680231, 0.42110308),
               (0.60637077, 0.27136769), (0.75157846, 0.70525231)]


def v2_empty():
    """No object anywhere, every box predicts (xy .5, wh = its anchor, conf c0): has_obj_mask = 0 kills the xy, wh,
    object-confidence and class terms (:96-104, :119-124), wh_pred/anchor = 1 kills the regulariser (:126-127),
    IoU with the empty truth box is 0 < ignore_thresh so no_obj_mask = 1 everywhere (:72-75):
        loss = w[2] * binary_weight * gh * gw * B * c0^2          (:112-117)"""
    N, g, A, C, c0 = 3, 13, 5, 20, 0.5
    yt = np.zeros((N, g, g, 5 + C), np.float32)
    yp = np.zeros((N, g, g, A, 5 + C), np.float32)
    yp[..., 0:2] = 0.5
    yp[..., 4] = c0
    yp[..., 5:] = 1.0 / C
    for b in range(A):
        yp[..., b, 2], yp[..., b, 3] = V2_ANCHORS5[b]
    kw = dict(binary_weight=0.5, loss_weight=[1, 1, 5, 1], ignore_thresh=0.6)
    expect = 5 * 0.5 * g * g * A * c0 ** 2          # = 528.125
    return dict(N=N, g=g, A=A, C=C, anchors=V2_ANCHORS5, yt=yt, yp=yp.reshape(N, g, g, -1), kw=kw, expect=expect)


def v2_one_object():
    """One cell, two anchors [[.2,.2],[.4,.4]], truth (.5,.5,.4,.4) class 0.
    box 0: (.1,.1,.2,.2) conf .3 -> IoU 0 with the truth, not responsible, no_obj (0 < .6)
    box 1: (.4,.7,.4 e^.5,.4 e^-.25) conf .8, class probs (.7,.3) -> IoU > 0: responsible (argmax, :63-65)"""
    anchors = [(0.2, 0.2), (0.4, 0.4)]
    t = (0.5, 0.5, 0.4, 0.4)
    b0 = (0.1, 0.1, 0.2, 0.2)
    b1 = (0.4, 0.7, 0.4 * math.exp(0.5), 0.4 * math.exp(-0.25))
    assert _iou_xywh(t, b1) > _iou_xywh(t, b0) == 0.0
    yt = np.zeros((1, 1, 1, 7), np.float32)
    yt[0, 0, 0] = [*t, 1, 1, 0]
    yp = np.zeros((1, 1, 1, 2, 7), np.float32)
    yp[0, 0, 0, 0] = [*b0, 0.3, 0.6, 0.4]
    yp[0, 0, 0, 1] = [*b1, 0.8, 0.7, 0.3]
    w, bw = [1.5, 1.2, 5, 0.8], 0.5
    scale = 2 - t[2] * t[3]                                              # box_loss_scale (:88)
    xy = scale * ((t[0] - b1[0]) ** 2 + (t[1] - b1[1]) ** 2)             # :90-96
    wh = scale * ((math.log(t[2] / 0.4) - 0.5) ** 2 + (math.log(t[3] / 0.4) + 0.25) ** 2)   # :82-86, :98-104
    c = (1 - 0.8) ** 2 + bw * (0.3 ** 2)                                  # :106-117
    p = -math.log(0.7)                                                    # :119-124 (p_true * log p_pred only)
    reg = 0.01 * (0.0 + 0.0 + 0.5 ** 2 + 0.25 ** 2)                       # :126-127 (box 0: wh = anchor)
    expect = w[0] * xy + w[1] * wh + w[2] * c + w[3] * p + reg
    kw = dict(binary_weight=bw, loss_weight=w, ignore_thresh=0.6)
    return dict(N=1, g=1, A=2, C=2, anchors=anchors, yt=yt, yp=yp.reshape(1, 1, 1, -1), kw=kw, expect=expect)


# ------------------------------------------------------------------------------------------------------------------
# YOLOv1.5 loss, yolov1_5/losses/loss.py:47-114 (confidence target = IoU, sqrt(wh), class term per cell)
# ------------------------------------------------------------------------------------------------------------------
def v1_one_object():
    t = (0.5, 0.5, 0.4, 0.4)
    b0 = (0.45, 0.55, 0.36, 0.49)     # responsible (IoU > 0)
    b1 = (0.1, 0.1, 0.2, 0.2)         # IoU 0
    iou0 = _iou_xywh(t, b0)
    assert iou0 > _iou_xywh(t, b1) == 0.0
    yt = np.zeros((1, 1, 1, 5 + 2), np.float32)
    yt[0, 0, 0] = [*t, 1, 1, 0]
    yp = np.zeros((1, 1, 1, 10 + 2), np.float32)
    yp[0, 0, 0] = [*b0, 0.9, *b1, 0.2, 0.8, 0.2]
    w, bw = [5, 5, 1, 1], 0.5
    xy = (t[0] - b0[0]) ** 2 + (t[1] - b0[1]) ** 2                                       # :70-75
    wh = (math.sqrt(t[2]) - math.sqrt(b0[2])) ** 2 + (math.sqrt(t[3]) - math.sqrt(b0[3])) ** 2   # :77-82
    c = (iou0 - 0.9) ** 2 + bw * (0.2 ** 2)                                               # :84-97 (box 1 is no_obj)
    p = -math.log(0.8)                                                                     # :103-107
    expect = w[0] * xy + w[1] * wh + w[2] * c + w[3] * p
    return dict(N=1, g=1, B=2, C=2, yt=yt, yp=yp, kw=dict(binary_weight=bw, loss_weight=w), expect=expect, iou=iou0)


# ------------------------------------------------------------------------------------------------------------------
# CIoU, yolov4/losses/loss.py:40-59, through the v4 loss with ONE anchor and ONE class (:109-166)
#   loss = w0 (1 - ciou) + w1 * [-(1-c)^gamma ln c] + w2 * [-ln p] + wh_reg_weight * sum log(wh/anchor)^2
# ------------------------------------------------------------------------------------------------------------------
def _ciou(t, p):
    iou = _iou_xywh(t, p)
    ex = max(t[0] + t[2] / 2, p[0] + p[2] / 2) - min(t[0] - t[2] / 2, p[0] - p[2] / 2)
    ey = max(t[1] + t[3] / 2, p[1] + p[3] / 2) - min(t[1] - t[3] / 2, p[1] - p[3] / 2)
    c2 = ex ** 2 + ey ** 2
    rho2 = (t[0] - p[0]) ** 2 + (t[1] - p[1]) ** 2
    v = 4.0 / math.pi ** 2 * (math.atan(t[2] / (t[3] + EPS)) - math.atan(p[2] / (p[3] + EPS))) ** 2
    alpha = v / (1 - iou + v)
    return iou - rho2 / c2 - alpha * v


CIOU_GEOMETRIES = {
    # name: (truth xywh, prediction xywh, closed form of the CIoU)
    "identical": ((0.5, 0.5, 0.4, 0.4), (0.5, 0.5, 0.4, 0.4), 0.16 / (0.16 + EPS)),           # rho = 0, v = 0
    "concentric_transposed": ((0.5, 0.5, 0.4, 0.2), (0.5, 0.5, 0.2, 0.4), None),              # rho = 0, v > 0
    "disjoint": ((0.25, 0.25, 0.2, 0.2), (0.75, 0.75, 0.2, 0.2), -0.5 / 0.98),                 # iou = 0, v = 0
}


def ciou_case(name):
    t, p, closed = CIOU_GEOMETRIES[name]
    ci = _ciou(t, p)
    if name == "concentric_transposed":
        # inter = .2*.2, union = .08+.08-.04 -> iou = 1/3; v = 4/pi^2 (atan 2 - atan .5)^2; alpha = v / (2/3 + v)
        v = 4 / math.pi ** 2 * (math.atan(2.0) - math.atan(0.5)) ** 2
        closed = 1 / 3 - (v / (2 / 3 + v)) * v
        assert abs(ci - closed) < 1e-6
    else:
        assert abs(ci - closed) < 1e-12
    yt = np.zeros((1, 1, 1, 6), np.float32)
    yt[0, 0, 0] = [*t, 1, 1]
    yp = np.zeros((1, 1, 1, 6), np.float32)
    yp[0, 0, 0] = [*p, 0.5, 0.5]
    w = [2.0, 5.0, 1.0]
    expect = w[0] * (1 - ci) + w[1] * (-(0.5 ** 2) * math.log(0.5)) + w[2] * (-math.log(0.5))   # wh = anchor: reg 0
    kw = dict(binary_weight=1, loss_weight=w, wh_reg_weight=0.01, ignore_thresh=0.6, truth_thresh=1, label_smooth=0,
              focal_loss_gamma=2)
    return dict(N=1, g=1, A=1, C=1, anchors=[(p[2], p[3])], yt=yt, yp=yp, kw=kw, expect=expect, ciou=ci)


# ------------------------------------------------------------------------------------------------------------------
# Metrics: yolov3/metrics/yolo_metrics.py:9-115 and yolov1_5/metrics/yolo_metrics.py:9-107
# grid 1 x 2: cell 0 holds an object (class 0), cell 1 is empty
# ------------------------------------------------------------------------------------------------------------------
def metrics_v3():
    yt = np.zeros((1, 1, 2, 7), np.float32)
    yt[0, 0, 0] = [0.5, 0.5, 0.4, 0.4, 1, 1, 0]
    yp = np.zeros((1, 1, 2, 2, 7), np.float32)
    yp[0, 0, 0, 0] = [0.5, 0.5, 0.4, 0.4, 0.9, 0.7, 0.3]     # IoU .16/(.16+eps), class 0 (right)
    yp[0, 0, 0, 1] = [0.5, 0.5, 0.2, 0.2, 0.4, 0.2, 0.8]     # IoU .25, class 1 (wrong)
    yp[0, 0, 1, 0] = [0.5, 0.5, 0.3, 0.3, 0.6, 0.5, 0.5]     # empty cell, confidence .6 > .5: objectness wrong
    yp[0, 0, 1, 1] = [0.5, 0.5, 0.3, 0.3, 0.3, 0.5, 0.5]
    iou = 0.16 / (0.16 + EPS)
    expect = dict(obj_acc=[1.0, 0.0],                         # binary_accuracy per cell (:24); Keras then means: .5
                  mean_iou=iou / (1 + EPS),                   # :45-49
                  class_acc=1.0 / (1 * 2 + EPS),              # :72-76: one of the two boxes of the object cell
                  recall=1.0 / (1 + EPS))                     # :106-113: max_b(iou * equal) = iou >= .5
    return dict(N=1, gh=1, gw=2, A=2, C=2, yt=yt, yp=yp.reshape(1, 1, 2, -1), expect=expect)


def metrics_v1():
    yt = np.zeros((1, 1, 2, 5 + 2), np.float32)
    yt[0, 0, 0] = [0.5, 0.5, 0.4, 0.4, 1, 0, 1]                                  # class 1
    yp = np.zeros((1, 1, 2, 10 + 2), np.float32)
    yp[0, 0, 0] = [0.5, 0.5, 0.2, 0.2, 0.4, 0.5, 0.5, 0.4, 0.4, 0.3, 0.2, 0.8]   # box 1 is the good one; class 1 right
    yp[0, 0, 1] = [0.5, 0.5, 0.3, 0.3, 0.2, 0.5, 0.5, 0.3, 0.3, 0.1, 0.9, 0.1]   # empty cell, max conf .2: right
    iou = 0.16 / (0.16 + EPS)
    expect = dict(obj_acc=[0.0, 1.0],                         # cell 0: max conf .4 <= .5 -> predicted "no object": wrong
                  mean_iou=iou / (1 + EPS), class_acc=1.0 / (1 + EPS), recall=1.0 / (1 + EPS))
    return dict(N=1, gh=1, gw=2, B=2, C=2, yt=yt, yp=yp, expect=expect)


# ------------------------------------------------------------------------------------------------------------------
# Keras layer semantics: padding of strided 'same' convs, ZeroPadding2D((1,0),(1,0)) + valid,
# tf.nn.space_to_depth channel order, -inf padding of 'same' max-pools
# ------------------------------------------------------------------------------------------------------------------
def keras_same_pad(size, k, s):
    """Keras/TF 'SAME': out = ceil(size / s); pad_total = max((out-1) s + k - size, 0); the SMALLER half goes first"""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2


def conv_tap_cases():
    """(H, W, k, stride, padding, tap (r, s)) with a one-hot filter: y[ho, wo] = x[ho*stride + r - pad_t, wo*stride + s - pad_l]
    or 0 outside. x[h, w] = 1 + h*W + w in channel 0. Expected maps are built from the padding rule alone."""
    cases = [
        (8, 8, 3, 2, "same", (0, 0)),        # even size: pad_total 1 -> top/left 0, bottom/right 1 (224 -> 112 pattern)
        (8, 8, 3, 2, "same", (2, 2)),        # the bottom/right zero row/column is visible at the last output
        (7, 7, 3, 2, "same", (0, 0)),        # odd size: pad_total 2 -> 1 / 1 (7 -> 4, the v1.5 grid)
        (7, 7, 3, 2, "same", (2, 2)),
        (28, 28, 7, 2, "same", (0, 0)),      # v1.5 stem: 7x7 stride 2: pad_total 5 -> top 2, bottom 3
        (28, 28, 7, 2, "same", (6, 6)),
        (9, 9, 3, 2, "darknet_s2", (0, 0)),  # ZeroPadding2D((1,0),(1,0)) + 'valid' (yolov3/models/backbone.py:31-34,61)
        (9, 9, 3, 2, "darknet_s2", (2, 2)),
        (6, 5, 3, 1, "same", (0, 2)),        # stride 1: symmetric
    ]
    out = []
    for H, W, k, s, pad, (r, q) in cases:
        if pad == "same":
            Ho, pt = keras_same_pad(H, k, s)
            Wo, pl = keras_same_pad(W, k, s)
        else:
            pt = pl = 1
            Ho, Wo = (H + 1 - k) // s + 1, (W + 1 - k) // s + 1
        x = (1 + np.arange(H * W, dtype=np.float32)).reshape(H, W)
        y = np.zeros((Ho, Wo), np.float32)
        for ho in range(Ho):
            for wo in range(Wo):
                hi, wi = ho * s + r - pt, wo * s + q - pl
                if 0 <= hi < H and 0 <= wi < W:
                    y[ho, wo] = x[hi, wi]
        out.append(dict(H=H, W=W, k=k, stride=s, padding=pad, tap=(r, q), x=x, y=y))
    return out


def space_to_depth_case():
    """tf.nn.space_to_depth(x, 2) (yolov2/models/darknet.py:52-56): out[h, w, (dy*2 + dx)*C + c] = x[2h+dy, 2w+dx, c].
    x[h, w, c] = 100 h + 10 w + c, C = 2, 4 x 4 -> 2 x 2 x 8"""
    x = np.zeros((1, 4, 4, 2), np.float32)
    for h in range(4):
        for w in range(4):
            for c in range(2):
                x[0, h, w, c] = 100 * h + 10 * w + c
    y = np.zeros((1, 2, 2, 8), np.float32)
    y[0, 0, 0] = [0, 1, 10, 11, 100, 101, 110, 111]
    y[0, 0, 1] = [20, 21, 30, 31, 120, 121, 130, 131]
    y[0, 1, 0] = [200, 201, 210, 211, 300, 301, 310, 311]
    y[0, 1, 1] = [220, 221, 230, 231, 320, 321, 330, 331]
    return x, y


def maxpool_cases():
    """'same' max-pools pad with -inf, never with 0 (all inputs negative: a zero pad would win every border maximum).
    x[h, w] = -(1 + h*W + w), so the maximum of a window is its top-left in-range element.
    (k, stride): 5/1, 9/1, 13/1 = SPP (yolov4/models/backbone.py:176-185), 2/1 = tiny-YOLOv3 (darknet.py:122), 2/2."""
    out = []
    for H, W, k, s in [(7, 6, 5, 1), (7, 6, 13, 1), (5, 5, 2, 1), (6, 6, 2, 2), (7, 7, 2, 2)]:
        Ho, pt = keras_same_pad(H, k, s)
        Wo, pl = keras_same_pad(W, k, s)
        x = -(1 + np.arange(H * W, dtype=np.float32)).reshape(H, W)
        y = np.zeros((Ho, Wo), np.float32)
        for ho in range(Ho):
            for wo in range(Wo):
                h0, w0 = max(ho * s - pt, 0), max(wo * s - pl, 0)
                y[ho, wo] = x[h0, w0]
        out.append(dict(H=H, W=W, k=k, stride=s, x=x, y=y, pad_t=pt, pad_l=pl))
    return out
