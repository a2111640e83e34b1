"""Static-graph executor for the YOLO training / inference hot path.

The reference builds tf.keras graphs (yolov3/models/*.py etc.) and lets Keras run
forward / autodiff / optimizer. Here a model is a flat list of units built once by
`GraphBuilder`; `Network.forward` / `Network.backward` walk it and enqueue the HIP kernels of
libyolo_hip.so on the current stream. Design points (MI355X-first):

  * NHWC fp32 activations, KRSC filters; all trainable parameters live in ONE flat buffer
    (`params`), with matching flat `grads`, Adam `m`/`v`: one optimizer launch per step and
    contiguous gradient buckets for the RCCL all-reduce (reverse construction order ==
    backward completion order).
  * every conv output and activation of a training step stays resident (288 GB HBM: no
    recompute, no offload); buffers are allocated once per batch size and reused.
  * conv+BN+activation(+residual Add) is one unit: conv kernel (BN statistics accumulated in
    fp64), a per-channel finalize, one fused normalise+activate(+add) pass.
  * gradient fan-out (residual skips, FPN taps) is resolved with "first writer aliases,
    later writers accumulate in place", so identity branches cost no copies.
"""
import math

import contextlib
import os

import numpy as np
import torch

from . import ops
from . import tape
from ._lib import ACT_LEAKY, ACT_LINEAR, ACT_MISH, YoloHipError

# words of the bounds / tickets block of a conv unit in Network._aux (zeroed at the start of every forward pass): [0] bound of the
# BatchNorm output, [1..68] the 68 words of yolo_bn_act_bwd_reduce_bound, [69..104] ticket words of its in-launch fold
AUX_WORDS = 112

ACT_NAMES = {ACT_LINEAR: "linear", ACT_LEAKY: "leaky", ACT_MISH: "mish"}


# --------------------------------------------------------------------------------------------
# parameter store
# --------------------------------------------------------------------------------------------
class ParamSpec:
    __slots__ = ("name", "shape", "offset", "size", "init", "trainable")

    def __init__(self, name, shape, offset, init, trainable):
        self.name, self.shape, self.offset = name, tuple(shape), offset
        self.size = int(np.prod(shape))
        self.init, self.trainable = init, trainable


class ParamStore:
    """Flat fp32 storage. Offsets are padded to 64 floats so every slice is 256-B aligned."""

    def __init__(self):
        self.specs = {}
        self.order = []
        self.total = 0

    def add(self, name, shape, init):
        if name in self.specs:
            raise ValueError(f"duplicate parameter {name}")
        spec = ParamSpec(name, shape, self.total, init, True)
        self.specs[name] = spec
        self.order.append(name)
        self.total += (spec.size + 63) // 64 * 64
        return spec

    def materialize(self, device, rng):
        host = np.zeros(self.total, dtype=np.float32)
        for name in self.order:
            s = self.specs[name]
            host[s.offset:s.offset + s.size] = s.init(rng, s.shape).reshape(-1)
        self.data = torch.from_numpy(host).to(device)
        return self.data

    def view(self, name):
        s = self.specs[name]
        return self.data[s.offset:s.offset + s.size]


def init_zeros(rng, shape):
    return np.zeros(shape, dtype=np.float32)


def init_ones(rng, shape):
    return np.ones(shape, dtype=np.float32)


def he_normal_krsc(rng, shape):
    """Keras he_normal: truncated normal (|z| <= 2), stddev = sqrt(2/fan_in)/0.87962566
    (SURVEY.md Appendix B); shape is KRSC so fan_in = kh*kw*Cin."""
    cout, kh, kw, cin = shape
    std = math.sqrt(2.0 / (kh * kw * cin)) / 0.87962566103423978
    z = rng.standard_normal(shape)
    bad = np.abs(z) > 2.0
    while bad.any():
        z[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(z) > 2.0
    return (z * std).astype(np.float32)


def random_normal_002(rng, shape):
    """yolov4/models/backbone.py:68: RandomNormal(mean=0.0, stddev=0.02)."""
    return (rng.standard_normal(shape) * 0.02).astype(np.float32)


# --------------------------------------------------------------------------------------------
# graph description
# --------------------------------------------------------------------------------------------
class TensorRef:
    """Symbolic activation: spatial shape without the batch dimension."""
    __slots__ = ("tid", "h", "w", "c", "name", "producer")

    def __init__(self, tid, h, w, c, name, producer=None):
        self.tid, self.h, self.w, self.c, self.name, self.producer = tid, h, w, c, name, producer

    @property
    def shape(self):
        return (None, self.h, self.w, self.c)


class Unit:
    kind = "unit"

    def __init__(self, name):
        self.name = name
        self.inputs = []
        self.out = None

    def param_names(self):
        return []


class ConvUnit(Unit):
    """conv [+bias] [+BatchNorm] [+activation] [+residual add]."""
    kind = "conv"

    def __init__(self, name, src, cout, k, stride, padding, bn, act, bias, residual, kernel_init):
        super().__init__(name)
        self.src, self.cout, self.k, self.stride, self.padding = src, cout, k, stride, padding
        self.bn, self.act, self.bias, self.residual = bn, act, bias, residual
        self.kernel_init = kernel_init
        self.inputs = [src] + ([residual] if residual is not None else [])


class HeadUnit(Unit):
    """All per-anchor 1x1 head convs of one output level fused into one [A*(5+C)] x Cin GEMM
    plus the per-channel activations (yolov3/models/__init__.py:40-65)."""
    kind = "head"

    def __init__(self, name, src, A, C, version, anchors, level, kernel_init):
        super().__init__(name)
        self.src, self.A, self.C, self.version, self.level = src, A, C, version, level
        self.kernel_init = kernel_init
        self.anchors = [tuple(map(float, a)) for a in anchors] if anchors is not None else None
        self.inputs = [src]


class UpsampleUnit(Unit):
    kind = "upsample"

    def __init__(self, name, src):
        super().__init__(name)
        self.src = src
        self.inputs = [src]


class ConcatUnit(Unit):
    kind = "concat"

    def __init__(self, name, srcs):
        super().__init__(name)
        self.srcs = list(srcs)
        self.inputs = list(srcs)


class MaxPoolUnit(Unit):
    kind = "maxpool"

    def __init__(self, name, src, k, stride, padding):
        super().__init__(name)
        self.src, self.k, self.stride, self.padding = src, k, stride, padding
        self.inputs = [src]


class SpaceToDepthUnit(Unit):
    kind = "space_to_depth"

    def __init__(self, name, src):
        super().__init__(name)
        self.src = src
        self.inputs = [src]


class GraphBuilder:
    """Functional-style builder, one call per reference layer group."""

    def __init__(self, input_shape, kernel_init=he_normal_krsc):
        h, w, c = input_shape
        self.tensors = []
        self.units = []
        self.kernel_init = kernel_init
        self.input = self._tensor(h, w, c, "input")
        self.outputs = []

    def _tensor(self, h, w, c, name, producer=None):
        t = TensorRef(len(self.tensors), h, w, c, name, producer)
        self.tensors.append(t)
        return t

    def _push(self, unit, h, w, c):
        unit.out = self._tensor(h, w, c, unit.name, unit)
        self.units.append(unit)
        return unit.out

    def conv(self, src, cout, k, name, stride=1, padding=None, bn=True, act=ACT_LEAKY, bias=False, residual=None,
             kernel_init=None):
        if padding is None:
            padding = "same"
        d = ops.conv_desc((1, src.h, src.w, src.c), cout, k, k, stride, padding)
        if residual is not None and (residual.h, residual.w, residual.c) != (d.Ho, d.Wo, cout):
            raise ValueError(f"{name}: residual shape mismatch")
        u = ConvUnit(name, src, cout, k, stride, padding, bn, act, bias, residual, kernel_init or self.kernel_init)
        return self._push(u, d.Ho, d.Wo, cout)

    def upsample(self, src, name):
        return self._push(UpsampleUnit(name, src), src.h * 2, src.w * 2, src.c)

    def concat(self, srcs, name):
        h, w = srcs[0].h, srcs[0].w
        for s in srcs:
            if (s.h, s.w) != (h, w):
                raise ValueError(f"{name}: concat of different spatial sizes")
        return self._push(ConcatUnit(name, srcs), h, w, sum(s.c for s in srcs))

    def maxpool(self, src, k, name, stride=None, padding="valid"):
        stride = stride or k
        if padding == "same":
            ho, wo = -(-src.h // stride), -(-src.w // stride)
        else:
            ho, wo = (src.h - k) // stride + 1, (src.w - k) // stride + 1
        return self._push(MaxPoolUnit(name, src, k, stride, padding), ho, wo, src.c)

    def space_to_depth(self, src, name):
        return self._push(SpaceToDepthUnit(name, src), src.h // 2, src.w // 2, src.c * 4)

    def head(self, src, A, C, version, anchors, name, level):
        cout = (5 * A + C) if version == 1 else A * (5 + C)
        out = self._push(HeadUnit(name, src, A, C, version, anchors, level, self.kernel_init), src.h, src.w, cout)
        self.outputs.append(out)
        return out


def count_params(builder):
    """(trainable, non_trainable) parameter counts of a built graph (no device needed)."""
    train = state = 0
    for u in builder.units:
        if u.kind == "conv":
            train += u.cout * u.k * u.k * u.src.c + (u.cout if u.bias else 0) + (2 * u.cout if u.bn else 0)
            state += 2 * u.cout if u.bn else 0
        elif u.kind == "head":
            train += u.out.c * u.src.c + u.out.c
    return train, state


def conv_flops_per_image(builder):
    """2*Ho*Wo*Cout*Cin*kh*kw summed over every conv (head convs included): SURVEY.md section 8d."""
    total = 0
    for u in builder.units:
        if u.kind == "conv":
            total += 2 * u.out.h * u.out.w * u.cout * u.src.c * u.k * u.k
        elif u.kind == "head":
            total += 2 * u.out.h * u.out.w * u.out.c * u.src.c
    return total


# --------------------------------------------------------------------------------------------
# runtime
# --------------------------------------------------------------------------------------------
class Network:
    def __init__(self, builder, device="cuda", seed=1234, unbiased_moving_var=True):
        if not torch.cuda.is_available():
            raise YoloHipError("tf2_yolo_amd needs a HIP device: there is no CPU fallback")
        self.device = torch.device(device)
        if self.device.index is not None and self.device.index != torch.cuda.current_device():
            raise YoloHipError(f"Network(device={device!r}): the kernels launch on the current device "
                               f"(cuda:{torch.cuda.current_device()}); call torch.cuda.set_device first "
                               "(one process per GPU)")
        self.tensors = builder.tensors
        self.units = builder.units
        self.input = builder.input
        self.outputs = list(builder.outputs)
        self.unbiased_moving_var = unbiased_moving_var
        self.params = ParamStore()   # trainable
        self.state = ParamStore()    # BN moving statistics (not trainable, not all-reduced)
        self._declare_params()
        rng = np.random.default_rng(seed)
        self.params.materialize(self.device, rng)
        self.state.materialize(self.device, rng)
        self.grads = torch.zeros_like(self.params.data)
        self._wT = torch.empty(self._wT_total, device=self.device, dtype=torch.float32)
        self._wT_valid = False
        self._overlap_wgrad = os.environ.get("YOLO_BWD_OVERLAP", "1") != "0"
        self._prep_beside = os.environ.get("YOLO_PREP_OVERLAP", "1") != "0"   # filter preparation beside the stem
        self._stem_fused = os.environ.get("YOLO_STEM_FUSED_BWD", "1") != "0"  # stem: BN backward apply + wgrad in one pass
        self._bn_tight = int(os.environ.get("YOLO_BN_TIGHT_BOUND", "0"))
        self._res_from_planes = os.environ.get("YOLO_RES_PLANES", "1") != "0"  # residual adds read the planes, not an fp32 copy
        self._wp_event = self._wT_event = None
        self._use_infer_graph = os.environ.get("YOLO_INFER_GRAPH", "1") != "0"
        self._fuse_infer = os.environ.get("YOLO_INFER_FUSE", "1") != "0"
        self._infer_graphs = {}
        self._wgrad_stream = None
        self._wgrad_pending = False
        self._bn_f32 = torch.zeros(max(self._bn_f32_total, 1), device=self.device, dtype=torch.float32)
        self._bn_f64 = torch.zeros(max(self._bn_f64_total, 1), device=self.device, dtype=torch.float64)
        # head anchors (v2-v4): views into ONE flat buffer with a gradient twin, so that trainable anchors
        # (yolov4/__init__.py:147-159) are one more small optimizer tensor
        self._anchors_dev, self._anchor_grad_views = {}, {}
        heads = [u for u in self.units if u.kind == "head" and u.anchors is not None]
        n_anch = sum(2 * len(u.anchors) for u in heads)
        self.anchors_flat = torch.zeros(max(n_anch, 1), device=self.device, dtype=torch.float32)
        self.anchor_grads = torch.zeros_like(self.anchors_flat)
        self.anchors_trainable = False
        off = 0
        for u in heads:
            n = 2 * len(u.anchors)
            self.anchors_flat[off:off + n] = torch.tensor(np.array(u.anchors, dtype=np.float32).reshape(-1))
            self._anchors_dev[u.name] = self.anchors_flat[off:off + n]
            self._anchor_grad_views[u.name] = self.anchor_grads[off:off + n]
            off += n
        self.has_anchors = n_anch > 0
        self.batch = None
        self.act = {}
        self.training = False
        self.grad_ready_hook = None   # called as hook(unit) after a unit's parameter grads are enqueued
        self.backward_begin_hook = None   # called at the start of backward (the gradient reducer's time origin)
        self._infer_scale_valid = False
        ops.create_side_streams(self.device)   # this device's filter-gradient / communication streams (fixed creation order)
        ops.ensure_conv_workspace()   # scratch of the persistent (stream-K) window kernel: small-batch launches
        ops.ensure_wgrad_workspace()  # slabs of the atomics-free filter / bias gradient reductions
        # consumers per tensor decide whether a tensor needs a gradient at all
        self._needs_grad = self._compute_needs_grad()
        self._n_consumers = {}
        for u in self.units:
            for t in u.inputs:
                self._n_consumers[t.tid] = self._n_consumers.get(t.tid, 0) + 1
        for t in self.outputs:
            self._n_consumers[t.tid] = self._n_consumers.get(t.tid, 0) + 1
        # the gradient of a Concatenate reaches the BatchNormalization backward of its sources as a channel SLICE of the
        # concat's gradient (ops.ChannelSlice: read in place with its row pitch) instead of being copied out per source
        self._concat_grad_slices = os.environ.get("YOLO_CONCAT_GRAD_SLICE", "1") != "0"
        self._dyp_per_layer = os.environ.get("YOLO_DYP_PER_LAYER", "1") != "0"
        # round 6: YOLO_BN_FOLD=1 finishes the BatchNormalization-backward reduction inside its own launch
        # (yolo_bn_act_bwd_reduce_fold_ld: the last workgroups to arrive fold the slots) instead of a bn_bwd_sum launch per
        # layer: 72 launches fewer. Default 0: measured neutral on YOLOv3-416 (28.99 / 28.99 ms) and 1 % SLOWER on YOLOv4-608
        # (37.20 -> 37.56: profiles/r06_a_bn_fold_ab_*.log) -- the two dependent memory-side round trips of the fold's tail
        # cost what the 8 us launch cost behind a queue that dispatches back to back
        self._bn_fold = os.environ.get("YOLO_BN_FOLD", "0") == "1"
        # The BatchNormalization-backward reduction of a conv + BN unit P (sums of dz and dz * xhat over dL/d(P.out)) is made
        # by the data gradient that COMPLETES dL/d(P.out) -- the last contribution in backward order, when that is a planes
        # data gradient (include/yolo_hip.h: yolo_conv2d_dgrad_planes_bnred): W.bnred_for = P on that writer W.
        # YOLO_BN_FUSED_REDUCE: 0 never, 1 every planes data gradient that completes a tensor, or a set of writer classes
        # "w" (3x3 stride-1 with enough tiles to run unsplit), "s" (stride 2), "p" (1x1), "h" (heads), "t" (the small
        # 3x3 layers that lose their split-K), e.g. "ws"
        # Default 0: measured (profiles/r05_a_*): the step does not get shorter with it -- the y tile costs the data gradient as
        # much as the standalone pass cost beside the filter-gradient stream.
        self._bn_fused_reduce = os.environ.get("YOLO_BN_FUSED_REDUCE", "0")
        self._plan_fused_reduce()
        # pre-split ("planes") copies of the filters for the LDS-DMA conv kernels: [Cout][taps*Cin] for
        # forward, [Cin][taps*Cout] for dgrad; refreshed when the parameters change
        wp = 0
        for u in self.units:
            if u.kind not in ("conv", "head"):
                continue
            cout = u.cout if u.kind == "conv" else u.out.c
            k = u.k if u.kind == "conv" else 1
            u.planes_fwd = ops.planes_fwd_ok(u.src.c, cout)
            u.planes_dgrad = ops.planes_dgrad_ok(u.src.c, cout) and self._needs_grad[u.src.tid]
            u.planes_wgrad = u.planes_fwd and ops.planes_wgrad_ok(u.src.c, cout, k * k, u.stride if u.kind == "conv" else 1)
            u.wp_off = u.wTp_off = -1
            # heads whose channel count is not a multiple of 16 (YOLOv3: 255): the backward kernels run on the
            # channel count rounded up to 16 - head gradient planes with zero columns, filters with zero rows
            u.cpad = 0
            if (u.kind == "head" and cout % 16 != 0 and not u.planes_dgrad
                    and os.environ.get("YOLO_HEAD_PAD", "1") != "0"):
                cp = (cout + 15) // 16 * 16
                if u.planes_fwd and ops.planes_wgrad_ok(u.src.c, cp, 1) and ops.planes_dgrad_ok(u.src.c, cp):
                    u.cpad = cp
                    u.wTp_pad_off, u.wTp_pad_bytes = wp, ops.planes_bytes(u.src.c, cp)
                    wp += (u.wTp_pad_bytes + 255) // 256 * 256
            if u.planes_fwd:
                u.wp_off, u.wp_bytes = wp, ops.planes_bytes(cout, k * k * u.src.c)
                wp += (u.wp_bytes + 255) // 256 * 256
            if u.planes_dgrad:
                u.wTp_off, u.wTp_bytes = wp, ops.planes_bytes(u.src.c, k * k * cout)
                wp += (u.wTp_bytes + 255) // 256 * 256
        self._wplanes = torch.empty(wp, device=self.device, dtype=torch.uint8) if wp else None
        for u in self.units:
            if u.kind == "head" and u.cpad:
                n = u.cpad * u.src.c
                u.w_pad = torch.zeros(n, device=self.device, dtype=torch.float32)    # [cpad][cin], zero rows past out.c
                u.wT_pad = torch.empty(n, device=self.device, dtype=torch.float32)   # [cin][cpad]
                u.dw_pad = torch.empty(n, device=self.device, dtype=torch.float32)   # filter-gradient scratch
        # fp32 copy of a BN/activation output is skipped (training) when every consumer is a conv that reads the
        # planes for both its forward and its filter gradient
        consumers = {}
        for u in self.units:
            for t in u.inputs:
                consumers.setdefault(t.tid, []).append(u)
        out_tids = {t.tid for t in self.outputs}
        for u in self.units:
            if u.kind == "conv":
                cs = consumers.get(u.out.tid, [])
                # the fp32 copy of a conv unit's output is written in training only if somebody reads it: consumers that take
                # it as a planes operand (planes conv / head units) do not, nor do conv + BN units that ADD it as their
                # residual (bn_act_fwd reads the residual from the planes too, yolo_bn_act_fwd_res_planes)
                def reads_planes_only(c):
                    if c.kind in ("conv", "head") and c.src.tid == u.out.tid and getattr(c, "residual", None) is not u.out:
                        return c.planes_fwd and c.planes_wgrad
                    return (self._res_from_planes and c.kind == "conv" and c.bn and getattr(c, "residual", None) is u.out
                            and c.src.tid != u.out.tid and c.cout % 16 == 0)
                has_planes_src = any(c.kind in ("conv", "head") and c.src.tid == u.out.tid and c.planes_fwd for c in cs)
                u.a_needed = ((not cs) or (u.out.tid in out_tids) or u.cout % 16 != 0 or not has_planes_src
                              or any(not reads_planes_only(c) for c in cs))
        # conv + BN + activation units whose ONLY consumer is a 2x2 / stride-2 pool that tiles their output (Darknet-19,
        # tiny-YOLOv3): in training the BatchNorm apply, the pool and the planes of the pooled tensor are ONE launch
        # (yolo_bn_act_maxpool2x2_fwd) and the unpooled activation is never written (YOLO_POOL_FUSE=0: three launches)
        self._pool_fuse = os.environ.get("YOLO_POOL_FUSE", "1") != "0"
        for u in self.units:
            u.pool_fuse = None
        if self._pool_fuse:
            for u in self.units:
                if u.kind != "conv" or not u.bn or u.residual is not None or u.out.tid in out_tids:
                    continue
                cs = consumers.get(u.out.tid, [])
                if len(cs) != 1 or cs[0].kind != "maxpool":
                    continue
                p = cs[0]
                if (p.k == 2 and p.stride == 2 and u.out.h % 2 == 0 and u.out.w % 2 == 0 and p.out.h * 2 == u.out.h
                        and p.out.w * 2 == u.out.w and u.cout % 8 == 0 and p.padding in ("same", "valid")):
                    pcs = consumers.get(p.out.tid, [])
                    p.f32_needed = (p.out.tid in out_tids) or (not pcs) or p.out.c % 16 != 0 or any(
                        not (c.kind in ("conv", "head") and c.src.tid == p.out.tid and c.planes_fwd and c.planes_wgrad
                             and getattr(c, "residual", None) is not p.out) for c in pcs)
                    u.pool_fuse = p
        # concat units write the planes of their result directly from the fp32 sources (yolo_split_planes_concat): the fp32
        # concatenation is produced only if somebody other than a planes convolution reads it; a source's bound is the one
        # its producer recorded -- BatchNorm's output bound, or, through pools / upsampling / space-to-depth (which cannot
        # increase max|x|), their input's
        self._concat_planes = os.environ.get("YOLO_CONCAT_PLANES", "1") != "0"
        self._bound_alias = {}
        for u in self.units:
            if u.kind in ("upsample", "maxpool", "space_to_depth"):
                self._bound_alias[u.out.tid] = self._bound_alias.get(u.src.tid, u.src.tid)
            elif u.kind == "concat":
                cs = consumers.get(u.out.tid, [])
                u.f32_needed = (u.out.tid in out_tids) or (not cs) or any(
                    not (c.kind in ("conv", "head") and c.src.tid == u.out.tid and c.planes_fwd and c.planes_wgrad
                         and getattr(c, "residual", None) is not u.out) for c in cs)
        self._wp_valid = False
        self._wTp_valid = False
        self._jobs_wp = self._jobs_wTp = self._jobs_wT = None
        # bounds for the planes scales (planes.hpp): 72 words per conv unit, zeroed every step; one float per
        # tensor = bound of that activation
        off = 0
        for u in self.units:
            u.aux_off = off       # [0] forward bound, [1..68] = the 68 words of yolo_bn_act_bwd_reduce_bound, [69..104] = the
            off += AUX_WORDS      # ticket words of its in-launch fold, [AUX_WORDS .. +C) = per-channel max|conv out| (conv epilogue)
            if u.kind == "conv" and u.bn:
                off += (u.cout + 7) // 8 * 8
        # (the per-tensor bounds live behind the units' words: zeroed with them at the start of every forward pass, which the
        # one-pass inference units need -- their kernels raise the bound of their result with atomicMax)
        ntb = max(len(self.tensors), 1) + 1
        self._aux = torch.zeros(max(off, 1) + ntb, device=self.device, dtype=torch.int32)
        self._tbound = self._aux[max(off, 1):].view(torch.float32)
        # A one-pass inference unit leaves the bound of its result as one word per workgroup of its last launch (their
        # maximum; include/yolo_hip.h: yolo_conv2d_fwd_infer_unit): _tword_n[tid] words of the tensor's row, allocated on first use
        self._twords = None
        self._tword_n = {}
        # {K, D} of every conv-BN unit for the a-priori bound of its inference output (ops.conv_pred_bound): made with
        # the folded scale / shift, i.e. once per set of weights
        self._pred = torch.zeros(2 * max(len(self.units), 1), device=self.device, dtype=torch.float32)
        for i, u in enumerate(self.units):
            u.pred_off = 2 * i
        self._infer_onepass = os.environ.get("YOLO_INFER_ONEPASS", "1") != "0"
        # round 6, bs-1 predict (no launch of the replayed graph costs less than ~4.5 us: DESIGN.md section 3.10):
        # (a) an UpSampling2D(2) whose only reader is a Concatenate is read THROUGH by the concat's planes pass (inference),
        # which also takes the sources' bounds as the words the one-pass units left instead of a fold launch per source
        # (yolo_split_planes_concat_ex); (b) a conv + BN unit nobody reads as planes (the FPN's lateral 1x1 in front of the
        # upsampling) still runs as a one-pass unit into planes of its own that nobody reads -- one launch instead of four;
        # (c) the heads: convolution + activation in one call (yolo_conv2d_fwd_head_unit). YOLO_INFER_SMALL_FUSE=0: off
        self._infer_small_fuse = os.environ.get("YOLO_INFER_SMALL_FUSE", "1") != "0"
        self._up_producer = {}
        for u in self.units:
            u.concat_reads_through = False
            if u.kind == "upsample" and self._n_consumers.get(u.out.tid, 0) == 1 and u.out.tid not in out_tids:
                cs = consumers.get(u.out.tid, [])
                if (len(cs) == 1 and cs[0].kind == "concat" and u.out.h == 2 * u.src.h and u.out.w == 2 * u.src.w
                        and u.out.c == u.src.c):
                    self._up_producer[u.out.tid] = u
        self._xplanes_infer = {}

    # ---- construction -------------------------------------------------------------------
    def _declare_params(self):
        wT = 0
        f32 = 0
        f64 = 0
        for u in self.units:
            if u.kind == "conv":
                cin = u.src.c
                u.p_kernel = self.params.add(f"{u.name}_conv/kernel", (u.cout, u.k, u.k, cin), u.kernel_init)
                u.p_bias = self.params.add(f"{u.name}_conv/bias", (u.cout,), init_zeros) if u.bias else None
                if u.bn:
                    u.p_gamma = self.params.add(f"{u.name}_bn/gamma", (u.cout,), init_ones)
                    u.p_beta = self.params.add(f"{u.name}_bn/beta", (u.cout,), init_zeros)
                    u.s_mean = self.state.add(f"{u.name}_bn/moving_mean", (u.cout,), init_zeros)
                    u.s_var = self.state.add(f"{u.name}_bn/moving_variance", (u.cout,), init_ones)
                    u.bn_f32_off = f32
                    f32 += 4 * u.cout            # scale, shift, save_mean, save_invstd
                    u.bn_f64_off = f64           # stats[SLOTS][2C]; all layers' statistics are contiguous: they are
                    f64 += ops.BN_STAT_SLOTS * 2 * u.cout   # the part of the fp64 buffer that is zeroed every step
                u.wT_off = wT
                wT += u.cout * u.k * u.k * cin
            elif u.kind == "head":
                cin = u.src.c
                cout = u.out.c
                u.p_kernel = self.params.add(f"{u.name}/kernel", (cout, 1, 1, cin), u.kernel_init)
                u.p_bias = self.params.add(f"{u.name}/bias", (cout,), init_zeros)
                u.wT_off = wT
                wT += cout * cin
        self._bn_stats_total = f64
        for u in self.units:              # red[RED_SLOTS+1][2C]: every slot is written before it is read, never zeroed
            if u.kind == "conv" and u.bn:
                u.bn_red_off = f64
                f64 += (ops.BN_RED_SLOTS + 1) * 2 * u.cout
        self._wT_total = wT
        self._bn_f32_total = f32
        self._bn_f64_total = f64

    def _plan_fused_reduce(self):
        """static walk of backward(): who contributes to dL/d(tensor), in which order. The last contributor of a tensor
        produced by a conv + BatchNormalization unit P gets W.bnred_for = P if it is a conv / head unit (whether its data
        gradient is a planes kernel is known later: _dgrad checks). A skipped writer (its own gradient never arrives) simply
        never sets the flag P's backward looks for, and P makes the reduction itself."""
        last = {}
        for u in reversed(self.units):
            u.bnred_for = None
            if u.kind == "conv":
                if u.bn and u.residual is not None and self._needs_grad[u.residual.tid]:
                    last[u.residual.tid] = ("add", u)
                if self._needs_grad[u.src.tid]:
                    last[u.src.tid] = ("dgrad", u)
            elif u.kind == "head":
                if self._needs_grad[u.src.tid]:
                    last[u.src.tid] = ("dgrad", u)
            else:
                for t in u.inputs:
                    if self._needs_grad[t.tid]:
                        last[t.tid] = ("other", u)
        mode = self._bn_fused_reduce
        if mode == "0":
            return
        for tid, (how, w) in last.items():
            p = self.tensors[tid].producer
            # (stride 2: the four parity classes of the data gradient must be ONE launch, csrc/conv.hip: dgrad_impl)
            k, st = (w.k, w.stride) if w.kind == "conv" else (1, 1)
            one_launch = st == 1 or (st == 2 and k == 3 and os.environ.get("YOLO_DGRAD_CLASSES", "1") != "0")
            if not (how == "dgrad" and one_launch and p is not None and p.kind == "conv" and p.bn and p.cout % 4 == 0):
                continue
            if w.kind == "head":
                cls = "h"
            elif st == 2:
                cls = "s"
            elif k == 1:
                cls = "p"
            else:
                cls = "w"
            w.bnred_class = cls
            if mode == "1" or cls in mode or (cls == "w" and "t" in mode):
                w.bnred_for = p

    def _compute_needs_grad(self):
        needs = {self.input.tid: False}
        for u in self.units:
            has_params = u.kind in ("conv", "head")
            needs[u.out.tid] = has_params or any(needs[t.tid] for t in u.inputs)
        return needs

    @property
    def num_params(self):
        return sum(self.params.specs[n].size for n in self.params.order)

    @property
    def num_state(self):
        return sum(self.state.specs[n].size for n in self.state.order)

    # ---- buffers ------------------------------------------------------------------------
    def allocate(self, N):
        if self.batch == N:
            return
        # every conv kernel addresses its operands through buffer descriptors with 32-bit offsets (the planes kernels check
        # it per launch, the register-staged ones do not): say so HERE, with the largest batch that fits, instead of
        # failing -- or wrapping around -- inside forward. 4 GiB per ACTIVATION is 192 images of YOLOv3-416 per GPU; a
        # larger global batch is what data parallelism is for.
        worst = max((t.h * t.w * t.c * 4 for t in self.tensors), default=0)
        if worst and N * worst + (1 << 20) >= (1 << 32):
            raise YoloHipError(f"batch {N}: the largest activation ({worst / 2 ** 20:.1f} MiB per image) would exceed the "
                               f"4 GiB a conv operand may span; the largest per-GPU batch for this model is "
                               f"{((1 << 32) - (1 << 20)) // worst - 1}")
        self.batch = N
        self.alloc_gen = getattr(self, "alloc_gen", 0) + 1   # captured step graphs (capture.py) belong to one allocation
        self._infer_graphs = {}   # captured graphs point into the buffers re-allocated below
        self.act = {}
        dev = self.device
        for u in self.units:
            oh, ow, oc = u.out.h, u.out.w, u.out.c
            if u.kind == "conv":
                u.desc = ops.conv_desc((N, u.src.h, u.src.w, u.src.c), u.cout, u.k, u.k, u.stride, u.padding)
                u.y = torch.empty((N, oh, ow, oc), device=dev, dtype=torch.float32)
                needs_a = u.bn or u.act != ACT_LINEAR
                u.a = torch.empty((N, oh, ow, oc), device=dev, dtype=torch.float32) if needs_a else u.y
                self.act[u.out.tid] = u.a
            elif u.kind == "head":
                u.desc = ops.conv_desc((N, u.src.h, u.src.w, u.src.c), oc, 1, 1, 1, "same")
                u.t = torch.empty((N, oh, ow, oc), device=dev, dtype=torch.float32)
                u.yact = torch.empty((N, oh, ow, oc), device=dev, dtype=torch.float32)
                self.act[u.out.tid] = u.yact
            elif u.kind == "maxpool":
                u.buf = torch.empty((N, oh, ow, oc), device=dev, dtype=torch.float32)
                u.argmax = torch.empty((N, oh, ow, oc), device=dev, dtype=torch.int32)
                if u.padding == "same":
                    _, u.pad_t = ops.same_pad(u.src.h, u.k, u.stride)
                    _, u.pad_l = ops.same_pad(u.src.w, u.k, u.stride)
                else:
                    u.pad_t = u.pad_l = 0
                self.act[u.out.tid] = u.buf
            else:
                u.buf = torch.empty((N, oh, ow, oc), device=dev, dtype=torch.float32)
                self.act[u.out.tid] = u.buf
        # slots of the BatchNormalization-backward reductions made by data gradients (_plan_fused_reduce)
        for u in self.units:
            u.bnred = None
        for w in self.units:
            p = getattr(w, "bnred_for", None)
            if p is None or not (w.planes_dgrad or (w.kind == "head" and w.cpad)):
                continue
            scale, shift, smean, sinv, _, _ = self._bn_bufs(p)
            if getattr(w, "bnred_class", "") == "w" and self._bn_fused_reduce != "1":
                # 3x3 stride-1 launches of at most 256 tiles run split-K (csrc/conv_win.hip: conv_split_parts), which the
                # fused form gives up: their own class "t"
                small = -(-N * w.src.h * w.src.w // 128) * -(-w.src.c // 128) <= 256
                if not (("t" if small else "w") in self._bn_fused_reduce):
                    continue
            cap = ops.bnred_slots_cap(w.desc)   # (a head with padded channels uses desc_pad: same pixels, same Cin)
            if cap <= 0:
                continue
            p.bnred = ops.BnReduce(p.y, scale, shift, smean, sinv, p.act,
                                   torch.empty(cap * 2 * p.cout, device=dev, dtype=torch.float32), cap,
                                   self._aux[p.aux_off + 1:p.aux_off + 69])
            p.bnred_done = False
        # planes of the activations that feed planes-capable convs (kept from forward for the filter
        # gradient), one scratch for the planes of the current layer's dy in backward
        self._xplanes = {}
        self._xplanes_infer = {}
        dyp = 0
        for u in self.units:
            if u.kind not in ("conv", "head"):
                continue
            if u.planes_fwd and u.src.tid not in self._xplanes:
                self._xplanes[u.src.tid] = torch.zeros(ops.planes_bytes(N * u.src.h * u.src.w, u.src.c), device=dev,
                                                       dtype=torch.uint8)
            if u.kind == "head" and u.cpad:
                dyp = max(dyp, ops.planes_bytes(N * u.out.h * u.out.w, u.cpad))
                u.desc_pad = ops.conv_desc((N, u.src.h, u.src.w, u.src.c), u.cpad, 1, 1, 1, "same")
            elif u.planes_dgrad or u.planes_wgrad:
                dyp = max(dyp, ops.planes_bytes(N * u.out.h * u.out.w, u.cout if u.kind == "conv" else u.out.c))
        # (two scratch buffers, used alternately: the filter gradient of layer L may still be reading its dy
        # planes on the second stream while layer L-1 produces its own)
        # (not allocated when every planes layer owns its buffer, the default below: _next_dyp_buffer creates them on demand)
        self._dyp_shared_bytes = dyp
        self._dyplanes2 = ([None, None] if self._dyp_per_layer else
                           [torch.empty(dyp, device=dev, dtype=torch.uint8) if dyp else None for _ in range(2)])
        self._dyp_events = [None, None]
        self._dyp_idx = 0
        # YOLO_DYP_PER_LAYER=1 (default): every layer has its OWN dy planes instead -- 288 GB of HBM make the reuse pointless
        # (YOLOv3-416 at bs 32: +4.6 GB), and the reuse costs the compute stream a wait on the filter-gradient stream's event
        # per layer: a barrier packet in its queue even when the event has long fired (2.4 us each, 75 per step, measured
        # with scripts/hip_probe/event_gap_probe.cpp: profiles/r05_a_event_gap_probe.log) plus the event record on the side
        # stream. Across steps the buffers are safe: backward joins the filter-gradient stream before it returns.
        self._dyp_own = {}
        if self._dyp_per_layer:
            for u in self.units:
                if u.kind == "head" and u.cpad:
                    nb = ops.planes_bytes(N * u.out.h * u.out.w, u.cpad)
                elif u.kind in ("conv", "head") and (u.planes_dgrad or u.planes_wgrad):
                    nb = ops.planes_bytes(N * u.out.h * u.out.w, u.cout if u.kind == "conv" else u.out.c)
                else:
                    continue
                self._dyp_own[u.name] = torch.empty(nb, device=dev, dtype=torch.uint8)
        self._xp_valid = set()

    def _tb_row(self, tid):
        if self._twords is None:
            self._twords = torch.zeros((max(len(self.tensors), 1) + 1) * ops.INFER_BOUND_WORDS, device=self.device,
                                       dtype=torch.int32)
        return self._twords[tid * ops.INFER_BOUND_WORDS:(tid + 1) * ops.INFER_BOUND_WORDS]

    def _tb_float(self, tid):
        """the recorded bound of tensor tid as ONE device float (folds the words of a one-pass inference unit once)"""
        n = self._tword_n.pop(tid, 0)
        if n:
            ops.fold_bound(self._tb_row(tid)[:n], self._tbound[tid:tid + 1])
        return self._tbound[tid:tid + 1]

    def _tb_words(self, tid):
        """the recorded bound of tensor tid as the one-pass units take it: the words such a unit left, or one float"""
        n = self._tword_n.get(tid, 0)
        return self._tb_row(tid)[:n] if n else self._tbound[tid:tid + 1]

    def _xp(self, t):
        """planes of activation tensor t for this forward pass (split once, shared by all consumers)"""
        buf = self._xplanes[t.tid]
        if t.tid not in self._xp_valid:
            ops.split_planes(self.act[t.tid], self.batch * t.h * t.w, t.c, out=buf)
            self._xp_valid.add(t.tid)
        return buf

    def _refresh_wplanes(self):
        if self._wp_valid or self._wplanes is None:
            return
        if self._jobs_wp is None:   # one batched launch for the planes of every filter (pointers never change)
            self._jobs_wp = ops.BatchJobs("split", self.device)
            for u in self.units:
                if u.kind in ("conv", "head") and u.planes_fwd:
                    cout = u.cout if u.kind == "conv" else u.out.c
                    k = u.k if u.kind == "conv" else 1
                    self._jobs_wp.add_split(self.params.view(u.p_kernel.name),
                                            self._wplanes[u.wp_off:u.wp_off + u.wp_bytes], cout, k * k * u.src.c)
        self._jobs_wp.run()
        self._wp_valid = True

    def _refresh_wTplanes(self):
        if self._wTp_valid or self._wplanes is None:
            return
        self._refresh_wplanes()   # bound donors of the transposed filters
        if self._jobs_wTp is None:
            self._jobs_wTp = ops.BatchJobs("split", self.device)
            for u in self.units:
                if u.kind == "head" and u.cpad:
                    self._jobs_wTp.add_split(u.wT_pad, self._wplanes[u.wTp_pad_off:u.wTp_pad_off + u.wTp_pad_bytes],
                                             u.src.c, u.cpad)
                elif u.kind in ("conv", "head") and u.planes_dgrad:
                    cout = u.cout if u.kind == "conv" else u.out.c
                    k = u.k if u.kind == "conv" else 1
                    n = cout * k * k * u.src.c
                    donor = self._wplanes[u.wp_off:u.wp_off + u.wp_bytes] if u.planes_fwd else None
                    self._jobs_wTp.add_split(self._wT[u.wT_off:u.wT_off + n],
                                             self._wplanes[u.wTp_off:u.wTp_off + u.wTp_bytes], u.src.c, k * k * cout,
                                             bound_from=donor)
        self._jobs_wTp.run()
        self._wTp_valid = True

    def _conv_fwd(self, u, xin, w, bias, out, stats=None):
        amax = self._aux[u.aux_off + AUX_WORDS:u.aux_off + AUX_WORDS + u.cout] if (stats is not None and self._tight_bound(u)) else None
        if u.planes_fwd:
            return ops.conv2d_fwd_planes(u.desc, self._xp(u.src), self._wplanes[u.wp_off:u.wp_off + u.wp_bytes], bias,
                                         out=out, stats=stats, absmax=amax)
        return ops.conv2d_fwd(u.desc, xin, w, bias, out=out, stats=stats, absmax=amax)

    def _tight_bound(self, u):
        """whether the conv epilogue records per-channel max|y| for the bound of the BatchNorm output (YOLO_BN_TIGHT_BOUND:
        0 never, 1 always, 2 = not for 1x1 layers). Without it bn_finalize takes |y - mean| <= sqrt(P var), sqrt(P) / (max / sigma)
        ~ 2^6 looser: the planes' absolute error floor moves from 2^-40 to 2^-34 of the bound (planes.hpp)."""
        m = self._bn_tight
        return m == 1 or (m == 2 and u.k != 1)

    def _bn_bufs(self, u):
        c = u.cout
        b = self._bn_f32[u.bn_f32_off:u.bn_f32_off + 4 * c]
        ns = ops.BN_STAT_SLOTS * 2 * c
        nr = (ops.BN_RED_SLOTS + 1) * 2 * c
        return (b[0:c], b[c:2 * c], b[2 * c:3 * c], b[3 * c:4 * c], self._bn_f64[u.bn_f64_off:u.bn_f64_off + ns],
                self._bn_f64[u.bn_red_off:u.bn_red_off + nr])

    # ---- inference through a captured HIP graph ------------------------------------------------------------
    def infer(self, x):
        """forward(x, training=False) replayed from a hipGraph: a batch-1 forward is ~230 launches of a few
        microseconds each, i.e. launch-bound when enqueued one by one. The graph is captured per batch size after an
        eager pass (which also refreshes the filter planes and the folded BN scales) and dropped whenever the
        parameters change. Outputs are the network's persistent head buffers, as with forward()."""
        if not self._use_infer_graph:
            return self.forward(x, training=False)
        x = x.contiguous()
        N = x.shape[0]
        g = self._infer_graphs.get(N)
        if g is None:
            self.forward(x, training=False)                      # eager: allocations, weight prep, lazy module init
            static_in = x.clone()
            self.forward(static_in, training=False)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                outs = self.forward(static_in, training=False)
            g = self._infer_graphs[N] = (graph, static_in, outs)
        graph, static_in, outs = g
        static_in.copy_(x)
        graph.replay()
        return outs

    def before_param_write(self):
        """Call BEFORE writing `params` / `state` on the current stream: a training-mode forward may still be
        transposing / splitting the old parameters on the second stream (its events are normally consumed by the first
        planes convolution and by backward; a forward without backward leaves them pending)."""
        for ev in (self._wp_event, self._wT_event):
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
        self._wp_event = self._wT_event = None

    def mark_params_changed(self):
        self.before_param_write()
        self._infer_graphs = {}
        self._wT_valid = False
        self._wp_valid = False
        self._wTp_valid = False
        self._infer_scale_valid = False

    # ---- forward ------------------------------------------------------------------------
    def forward(self, x, training=False, stop_after=None):
        """x: float32 CUDA tensor [N,H,W,C]. Returns the list of head outputs (coarse -> fine).
        stop_after (benchmarks only: bench.py's Darknet-53 block): name of the unit behind which the pass ends; returns None."""
        if x.dtype != torch.float32 or not x.is_cuda:
            raise YoloHipError("forward expects a float32 CUDA tensor")
        x = x.contiguous()
        N = x.shape[0]
        if tuple(x.shape[1:]) != (self.input.h, self.input.w, self.input.c):
            raise YoloHipError(f"input shape {tuple(x.shape[1:])} != model input "
                               f"{(self.input.h, self.input.w, self.input.c)}")
        self.allocate(N)
        self.training = training
        self.act[self.input.tid] = x
        self._xp_valid = set()
        self._tbound_set = set()
        self._tword_n = {}        # tensors whose recorded bound is (still) the words of a one-pass inference unit
        # (a pending event of an earlier forward stays until a planes conv / a backward has waited for it)
        if training and self._overlap_wgrad and self._prep_beside and not (self._wp_valid and self._wT_valid and self._wTp_valid):
            # the filters' planes (needed by the first planes conv) and their transposed forms (needed by backward) are
            # made on the second stream while the stem runs: 0.45 ms of HBM-bound launches beside MFMA-bound ones
            if self._wgrad_stream is None:
                self._wgrad_stream = ops.concurrent_stream("wgrad", device=self.device)
            side = self._wgrad_stream
            tape.wait_stream(side, torch.cuda.current_stream())
            with torch.cuda.stream(side):
                self._refresh_wplanes()
                self._wp_event = tape.record_event(side)
                self._refresh_wT()
                self._refresh_wTplanes()
                self._wT_event = tape.record_event(side)
        else:
            self._refresh_wplanes()
        ops.zero_bytes(self._aux)   # bounds / per-channel maxima of this pass
        if training:
            ops.zero_bytes(self._bn_f64[:self._bn_stats_total])
            self._infer_scale_valid = False  # moving statistics (and the shared scale/shift) change
            # captured inference graphs were recorded with the folded scale/shift valid (no bn_fold_inference
            # inside): a training-mode forward overwrites those buffers with batch statistics, so drop them
            self._infer_graphs = {}
        P = self.params
        prev_name = None
        for u in self.units:
            if stop_after is not None and prev_name == stop_after:
                self.training = False   # an incomplete pass: backward() must not run on it
                return None
            prev_name = getattr(u, "name", None)
            if self._wp_event is not None and u.kind in ("conv", "head") and u.planes_fwd:
                tape.wait_event(torch.cuda.current_stream(), self._wp_event)
                self._wp_event = None
            if u.kind == "conv":
                xin = self.act[u.src.tid]
                w = P.view(u.p_kernel.name)
                bias = P.view(u.p_bias.name) if u.p_bias is not None else None
                if u.bn:
                    scale, shift, smean, sinv, stats, _ = self._bn_bufs(u)
                    gamma, beta = P.view(u.p_gamma.name), P.view(u.p_beta.name)
                    if training:
                        # a conv bias in front of BatchNormalization (YOLOv1.5 / v2) cancels in (y - mean): the convolution
                        # runs WITHOUT it -- u.y, the statistics and the saved mean are those of the bias-free tensor, so a
                        # large bias cannot eat the digits of var = E[y^2] - mean^2 -- and only the moving mean adds it back
                        self._conv_fwd(u, xin, w, None, u.y, stats)
                        ops.bn_finalize(stats, u.y.numel() // u.cout, u.cout, gamma, beta,
                                        self.state.view(u.s_mean.name), self.state.view(u.s_var.name),
                                        scale, shift, smean, sinv, unbiased=self.unbiased_moving_var,
                                        bound=self._aux[u.aux_off:u.aux_off + 1],
                                        absmax=(self._aux[u.aux_off + AUX_WORDS:u.aux_off + AUX_WORDS + u.cout]
                                                if self._tight_bound(u) else None), mean_offset=bias)
                    elif self._fuse_infer and self._fused_infer_unit(u, bias, gamma, beta, scale, shift):
                        continue
                    elif self._fuse_infer and self._infer_onepass and self._stem_infer_unit(u, xin, w, bias, gamma, beta, scale, shift):
                        continue
                    else:
                        amax = self._aux[u.aux_off + AUX_WORDS:u.aux_off + AUX_WORDS + u.cout]
                        if u.planes_fwd:
                            ops.conv2d_fwd_planes(u.desc, self._xp(u.src), self._wplanes[u.wp_off:u.wp_off + u.wp_bytes],
                                                  bias, out=u.y, absmax=amax)
                        else:
                            ops.conv2d_fwd(u.desc, xin, w, bias, out=u.y, absmax=amax)
                        if not self._infer_scale_valid:
                            ops.bn_fold_inference(u.cout, gamma, beta, self.state.view(u.s_mean.name),
                                                  self.state.view(u.s_var.name), scale, shift)
                        ops.bn_infer_bound(u.cout, scale, shift, amax, self._aux[u.aux_off:u.aux_off + 1])
                    if training and u.pool_fuse is not None:
                        # BatchNorm apply + activation + the 2x2 pool behind it + the pooled tensor's planes in one launch; the
                        # unpooled activation is not written (its only reader would have been the pool)
                        pu = u.pool_fuse
                        ppl = self._xplanes.get(pu.out.tid) if pu.out.c % 16 == 0 else None
                        tb = self._tbound
                        ops.bn_act_maxpool2x2_fwd(u.y, u.cout, scale, shift, u.act, pu.argmax,
                                                  out=pu.buf if (pu.f32_needed or ppl is None) else None, planes=ppl,
                                                  bn_bound=self._aux[u.aux_off:u.aux_off + 1],
                                                  out_bound=tb[pu.out.tid:pu.out.tid + 1])
                        self._tbound_set.add(pu.out.tid)
                        if ppl is not None:
                            self._xp_valid.add(pu.out.tid)
                        pu.fused_this_pass = True
                        continue
                    res = self.act[u.residual.tid] if u.residual is not None else None
                    res_pl = None
                    if (res is not None and training and self._res_from_planes and u.cout % 16 == 0
                            and u.residual.tid in self._xp_valid and u.residual.tid in self._tbound_set):
                        res_pl, res = self._xplanes[u.residual.tid], None   # (its fp32 copy may not have been written)
                    # consumers that are planes-capable convs get their operand straight from this kernel; the scale
                    # comes from the bound bn_finalize (training) / bn_infer_bound (inference) derived from the conv
                    # epilogue's per-channel max|y|
                    pl = self._xplanes.get(u.out.tid) if u.cout % 16 == 0 else None
                    if u.residual is not None and u.residual.tid not in self._tbound_set:
                        pl = None   # residual without a recorded bound: split its consumers' operand separately
                    else:
                        self._tbound_set.add(u.out.tid)
                    tb = self._tbound
                    ops.bn_act_fwd(u.y, u.cout, scale, shift, u.act, res, out=u.a, planes=pl,
                                   want_out=not (training and pl is not None and not u.a_needed),
                                   bn_bound=self._aux[u.aux_off:u.aux_off + 1],
                                   residual_bound=self._tb_float(u.residual.tid) if res is not None else None,
                                   out_bound=tb[u.out.tid:u.out.tid + 1], residual_planes=res_pl)
                    if pl is not None:
                        self._xp_valid.add(u.out.tid)
                else:
                    self._conv_fwd(u, xin, w, bias, u.y)
                    if u.act != ACT_LINEAR:
                        ops.act_fwd(u.y, u.act, out=u.a)
                    if u.residual is not None:
                        raise YoloHipError("residual without BN is not used by any reference graph")
            elif u.kind == "head":
                xin = self.act[u.src.tid]
                if not training and self._infer_small_fuse and self._fuse_infer and u.planes_fwd:
                    ops.conv2d_fwd_head_unit(u.desc, self._xp(u.src), self._wplanes[u.wp_off:u.wp_off + u.wp_bytes],
                                             P.view(u.p_bias.name), u.A, u.C, u.version, self._anchors_dev.get(u.name),
                                             u.t, u.yact)
                    continue
                self._conv_fwd(u, xin, P.view(u.p_kernel.name), P.view(u.p_bias.name), u.t)
                ops.head_act_fwd(u.t, u.A, u.C, u.version, self._anchors_dev.get(u.name), out=u.yact)
            elif u.kind == "upsample":
                if (not training and self._infer_small_fuse and self._concat_planes
                        and u.out.tid in self._up_producer):
                    u.concat_reads_through = True   # (the Concatenate behind it reads the half-size tensor itself)
                    continue
                ops.upsample2x_fwd(self.act[u.src.tid], u.buf, u.out.c, 0)
            elif u.kind == "concat":
                pl = self._xplanes.get(u.out.tid) if self._concat_planes else None
                slots = [self._bound_alias.get(s.tid, s.tid) for s in u.srcs]
                ups = [self._up_producer.get(s.tid) for s in u.srcs]
                ups = [p if (p is not None and p.concat_reads_through) else None for p in ups]
                if (pl is not None and len(u.srcs) <= 4 and all(sl in self._tbound_set for sl in slots)
                        and all(s.c % 8 == 0 for s in u.srcs)):
                    tb = self._tbound
                    if any(p is not None for p in ups):
                        ops.split_planes_concat_ex([self.act[p.src.tid] if p is not None else self.act[s.tid]
                                                    for s, p in zip(u.srcs, ups)], [s.c for s in u.srcs],
                                                   [self._tb_words(sl) for sl in slots], self.batch * u.out.h * u.out.w, pl,
                                                   upsample=[p is not None for p in ups], hw=(u.out.h, u.out.w),
                                                   dst32=u.buf if u.f32_needed else None,
                                                   out_bound=tb[u.out.tid:u.out.tid + 1])
                        for p in ups:
                            if p is not None:
                                p.concat_reads_through = False
                    else:
                        ops.split_planes_concat([self.act[s.tid] for s in u.srcs], [s.c for s in u.srcs],
                                                [self._tb_float(sl) for sl in slots], self.batch * u.out.h * u.out.w, pl,
                                                dst32=u.buf if u.f32_needed else None,
                                                out_bound=tb[u.out.tid:u.out.tid + 1])
                    self._xp_valid.add(u.out.tid)
                    self._tbound_set.add(u.out.tid)
                    continue
                for p in ups:   # (this Concatenate cannot read through after all: make the upsampled tensors now)
                    if p is not None:
                        ops.upsample2x_fwd(self.act[p.src.tid], p.buf, p.out.c, 0)
                        p.concat_reads_through = False
                off = 0
                for s in u.srcs:
                    ops.copy_channels_in(self.act[s.tid], s.c, u.buf, u.out.c, off)
                    off += s.c
            elif u.kind == "maxpool":
                if getattr(u, "fused_this_pass", False):   # made by its producer's BatchNorm launch (pool_fuse)
                    u.fused_this_pass = False
                    continue
                ops.maxpool_fwd(self.act[u.src.tid], u.k, u.stride, u.pad_t, u.pad_l, u.out.h, u.out.w, u.buf,
                                u.out.c, 0, u.argmax if training else None)
            elif u.kind == "space_to_depth":
                ops.space_to_depth2_fwd(self.act[u.src.tid], u.buf, u.out.c, 0)
            else:
                raise YoloHipError(f"unknown unit kind {u.kind}")
        if not training:
            self._infer_scale_valid = True
        return [self.act[t.tid] for t in self.outputs]

    def _stem_infer_unit(self, u, xin, w, bias, gamma, beta, scale, shift):
        """The RGB stem's inference unit in one launch (yolo_stem_fwd_infer_unit): direct fp32 convolution, folded
        BatchNormalization + activation in registers, the planes of the next convolution written by the same kernel; the image's
        max|x| comes from yolo_absmax_words. Returns False for any other unit."""
        pl = self._xplanes.get(u.out.tid)
        if u.residual is not None or pl is None or u.src.tid != self.input.tid or not ops.stem_infer_supported(u.desc):
            return False
        if not self._infer_scale_valid:
            ops.bn_fold_inference(u.cout, gamma, beta, self.state.view(u.s_mean.name), self.state.view(u.s_var.name),
                                  scale, shift)
            if getattr(u, "stem_wt", None) is None:
                u.stem_wt = torch.empty(28 * 32, device=self.device, dtype=torch.float32)
            ops.stem_filter_prep(w, bias, u.stem_wt)
            pred = self._pred[u.pred_off:u.pred_off + 2]
            ops.zero_bytes(pred)
            ops.conv_pred_bound(w, u.cout, 27, scale, shift, bias, pred)
        epi = {ACT_LEAKY: ops.EPI_AFFINE_LEAKY, ACT_MISH: ops.EPI_AFFINE_MISH}.get(u.act, ops.EPI_AFFINE)
        row = self._tb_row(self.input.tid)
        n_in = ops.absmax_words(xin, row)
        nw = ops.stem_fwd_infer_unit(u.desc, xin, u.stem_wt, epi, scale, shift, self._pred[u.pred_off:u.pred_off + 2], row[:n_in],
                                     u.a if u.a_needed else None, pl, self._tb_row(u.out.tid))
        self._tword_n[u.out.tid] = nw
        self._tbound_set.add(u.out.tid)
        self._xp_valid.add(u.out.tid)
        return True

    def _fused_infer_unit(self, u, bias, gamma, beta, scale, shift):
        """Inference form of a conv-BN-activation(-Add) unit in two launches (include/yolo_hip.h, fused epilogue): the
        folded BatchNormalization, the activation and the residual Add run in the conv's epilogue, then ONE pass turns
        the result into the planes of the consumer convolutions (bound = the epilogue's per-channel max + the
        residual's bound). Returns False when the unit keeps the three-launch path (no planes operand / consumer)."""
        if not u.planes_fwd or u.cout % 16 != 0:
            return False
        pl = self._xplanes.get(u.out.tid)
        own_planes = pl is not None
        if pl is None and self._infer_small_fuse and self._infer_onepass:
            # nobody reads this unit's result as planes (the FPN's lateral 1x1 conv in front of UpSampling2D): at few
            # output pixels the one-pass unit is still ONE launch where conv + reduce + bound + BatchNorm apply are four --
            # it writes planes of its own that nobody reads
            rows = self.batch * u.out.h * u.out.w
            if rows <= 4096:
                pl = self._xplanes_infer.get(u.out.tid)
                if pl is None:
                    pl = self._xplanes_infer[u.out.tid] = torch.zeros(ops.planes_bytes(rows, u.cout), device=self.device,
                                                                     dtype=torch.uint8)
        if pl is None or (u.residual is not None and u.residual.tid not in self._tbound_set):
            return False
        if not self._infer_scale_valid:
            ops.bn_fold_inference(u.cout, gamma, beta, self.state.view(u.s_mean.name), self.state.view(u.s_var.name),
                                  scale, shift)
        amax = self._aux[u.aux_off + AUX_WORDS:u.aux_off + AUX_WORDS + u.cout]
        res = self.act[u.residual.tid] if u.residual is not None else None
        epi = {ACT_LEAKY: ops.EPI_AFFINE_LEAKY, ACT_MISH: ops.EPI_AFFINE_MISH}.get(u.act, ops.EPI_AFFINE)
        tb = self._tbound
        src_slot = self._bound_alias.get(u.src.tid, u.src.tid)
        if self._infer_onepass and src_slot in self._tbound_set:
            # one pass: the kernel that finishes the tile writes the planes too, scaled by an a-priori bound
            pred = self._pred[u.pred_off:u.pred_off + 2]
            if not self._infer_scale_valid:
                ops.zero_bytes(pred)
                ops.conv_pred_bound(self.params.view(u.p_kernel.name), u.cout, u.k * u.k * u.src.c, scale, shift, bias, pred)
            nw = ops.conv2d_fwd_infer_unit(u.desc, self._xp(u.src), self._wplanes[u.wp_off:u.wp_off + u.wp_bytes], bias,
                                           epi, scale, shift, res, u.a, amax, pred, self._tb_words(src_slot),
                                           self._tb_words(u.residual.tid) if u.residual is not None else None, pl,
                                           self._tb_row(u.out.tid), tb[u.out.tid:u.out.tid + 1])
            if nw:
                self._tword_n[u.out.tid] = nw
            self._tbound_set.add(u.out.tid)
            if own_planes:
                self._xp_valid.add(u.out.tid)
            return True
        if not own_planes:
            return False
        ops.conv2d_fwd_planes_epi(u.desc, self._xp(u.src), self._wplanes[u.wp_off:u.wp_off + u.wp_bytes], bias, epi, scale,
                                  shift, residual=res, out=u.a, absmax=amax)
        ops.split_planes_absmax(u.a, self.batch * u.out.h * u.out.w, u.cout, amax, pl,
                                extra_bound=self._tb_float(u.residual.tid) if u.residual is not None else None,
                                out_bound=tb[u.out.tid:u.out.tid + 1])
        self._tbound_set.add(u.out.tid)
        self._xp_valid.add(u.out.tid)
        return True

    # ---- backward -----------------------------------------------------------------------
    def _takes_grad_slice(self, u):
        """can unit u's backward read dL/d(out) as a channel slice of a wider tensor? conv + BatchNormalization units without a
        residual Add (bn_act_bwd reads dout twice and nothing else does), not the fused stem path"""
        if u.kind != "conv" or not u.bn or u.residual is not None or u.cout % 8 != 0:
            return False
        need_pl = u.planes_wgrad or u.planes_dgrad
        if self._stem_fused and not need_pl and not self._needs_grad[u.src.tid] and ops.stem_bn_bwd_supported(u.desc):
            return False
        return True

    def _add_grad(self, grads, t, buf):
        """Contribute `buf` (same shape as tensor t) to dL/dt: alias if first, else add in place."""
        if not self._needs_grad[t.tid]:
            return
        cur = grads.get(t.tid)
        if cur is None:
            grads[t.tid] = buf
        else:
            ops.axpy(cur, buf)

    def _refresh_wT(self):
        if self._wT_valid:
            return
        if self._jobs_wT is None:
            self._jobs_wT = ops.BatchJobs("transpose", self.device)
            for u in self.units:
                if u.kind == "head" and u.cpad:
                    self._jobs_wT.add_transpose(u.w_pad, u.wT_pad, u.cpad, 1, u.src.c)
                elif u.kind in ("conv", "head"):
                    cout = u.cout if u.kind == "conv" else u.out.c
                    taps = u.k * u.k if u.kind == "conv" else 1
                    n = cout * taps * u.src.c
                    self._jobs_wT.add_transpose(self.params.view(u.p_kernel.name), self._wT[u.wT_off:u.wT_off + n],
                                                cout, taps, u.src.c)
        for u in self.units:
            if u.kind == "head" and u.cpad:
                tape.torch_op(lambda u=u: u.w_pad[:u.out.c * u.src.c].copy_(self.params.view(u.p_kernel.name).reshape(-1)))
        self._jobs_wT.run()
        self._wT_valid = True

    def _gview(self, spec):
        return self.grads[spec.offset:spec.offset + spec.size]

    def backward(self, douts):
        """douts: list of dL/d(output) tensors (one per head, same order as forward's result).
        Parameter gradients are ACCUMULATED into self.grads (zeroed by the optimizer step)."""
        if not self.training:
            raise YoloHipError("backward() requires a preceding forward(training=True)")
        if self.backward_begin_hook is not None:
            tape.host_call(self.backward_begin_hook)
        if self._wT_event is not None:
            tape.wait_event(torch.cuda.current_stream(), self._wT_event)
            self._wT_event = None
        self._refresh_wT()
        self._refresh_wTplanes()
        for u in self.units:
            if getattr(u, "bnred", None) is not None:
                u.bnred_done = False
        grads = {}
        for t, g in zip(self.outputs, douts):
            grads[t.tid] = g
        N = self.batch
        for u in reversed(self.units):
            dout = grads.pop(u.out.tid, None)
            if dout is None:
                continue  # output unused by the loss
            if isinstance(dout, ops.ChannelSlice) and not self._takes_grad_slice(u):
                dout = dout.dense((N, u.out.h, u.out.w, u.out.c))
            if u.kind == "conv":
                xin = self.act[u.src.tid]
                if u.bn:
                    scale, shift, smean, sinv, _, red = self._bn_bufs(u)
                    if u.residual is not None:
                        self._add_grad(grads, u.residual, dout)
                    # the gradient of the conv output goes straight into the planes the filter / data gradient
                    # kernels read; the fp32 copy is written only if an fp32-path kernel still needs it
                    need_pl = u.planes_wgrad or u.planes_dgrad
                    # (a conv bias in front of BatchNormalization has the exact gradient 0 -- it cancels in y - mean -- so
                    # nothing is computed for it: its slice of `grads` stays zero and no fp32 dy is needed on its account)
                    need_f32 = (not u.planes_wgrad) or (self._needs_grad[u.src.tid] and not u.planes_dgrad)
                    if (self._stem_fused and not need_pl and not self._needs_grad[u.src.tid]
                            and ops.stem_bn_bwd_supported(u.desc)):
                        # the stem: BN / activation backward apply + filter gradient in one pass, no 32-channel dy tensor
                        ops.stem_bn_bwd_wgrad(u.desc, xin, u.y, dout, scale, shift, smean, sinv, u.act, red,
                                              self._gview(u.p_gamma), self._gview(u.p_beta), self._gview(u.p_kernel),
                                              fused=self._take_fused(u, dout))
                        if self.grad_ready_hook is not None:
                            tape.host_call(lambda u=u: self.grad_ready_hook(u))
                        continue
                    dyp = self._next_dyp_buffer(u) if need_pl else None
                    dy = ops.bn_act_bwd(u.y, dout, u.cout, self.params.view(u.p_gamma.name), scale, shift, smean,
                                        sinv, u.act, red, self._gview(u.p_gamma), self._gview(u.p_beta),
                                        planes=dyp, want_dx=need_f32,
                                        bound_aux=self._aux[u.aux_off + 1:u.aux_off + 69], fused=self._take_fused(u, dout),
                                        tickets=(self._aux[u.aux_off + 69:u.aux_off + 69 + ops.BN_FOLD_TICKET_WORDS]
                                                 if self._bn_fold else None))
                else:
                    dy = ops.act_bwd(u.y, dout, u.act) if u.act != ACT_LINEAR else dout
                    dyp = self._dyp(u, dy)
                dbias = self._gview(u.p_bias) if (u.p_bias is not None and not u.bn) else None
                with self._beside_backward(dy):
                    if u.planes_wgrad:
                        ops.conv2d_wgrad_planes(u.desc, self._xplanes[u.src.tid], dyp, self._gview(u.p_kernel), dy=dy,
                                                dbias=dbias)
                    else:
                        ops.conv2d_wgrad(u.desc, xin, dy, self._gview(u.p_kernel), dbias)
                self._dgrad(grads, u, dy, u.cout * u.k * u.k * u.src.c, dyp)
            elif u.kind == "head":
                xin = self.act[u.src.tid]
                dt = ops.head_act_bwd(u.yact, dout, u.A, u.C, u.version, self._anchors_dev.get(u.name),
                                      danchors=self._anchor_grad_views.get(u.name) if self.anchors_trainable else None)
                if u.cpad:
                    rows = N * u.out.h * u.out.w
                    dtp = ops.split_planes_padded(dt, rows, u.out.c, out=self._next_dyp_buffer(u))
                    with self._beside_backward(dt):
                        ops.zero_bytes(u.dw_pad)
                        ops.conv2d_wgrad_planes(u.desc_pad, self._xplanes[u.src.tid], dtp, u.dw_pad)
                        ops.axpy(self._gview(u.p_kernel), u.dw_pad[:u.out.c * u.src.c])
                        ops.conv2d_wgrad_bias(dt, rows, u.out.c, self._gview(u.p_bias))
                    if self._needs_grad[u.src.tid]:
                        wTp = self._wplanes[u.wTp_pad_off:u.wTp_pad_off + u.wTp_pad_bytes]
                        cur = grads.get(u.src.tid)
                        b = self._bnred_of(u, cur)
                        if cur is None:
                            grads[u.src.tid] = ops.conv2d_dgrad_planes(u.desc_pad, dtp, wTp, bnred=b)
                        else:
                            ops.conv2d_dgrad_planes(u.desc_pad, dtp, wTp, dx=cur, accumulate=True, bnred=b)
                    if self.grad_ready_hook is not None:
                        tape.host_call(lambda u=u: self.grad_ready_hook(u))
                    continue
                dtp = self._dyp(u, dt)
                with self._beside_backward(dt):
                    if u.planes_wgrad:
                        ops.conv2d_wgrad_planes(u.desc, self._xplanes[u.src.tid], dtp, self._gview(u.p_kernel), dy=dt,
                                                dbias=self._gview(u.p_bias))
                    else:
                        ops.conv2d_wgrad(u.desc, xin, dt, self._gview(u.p_kernel), self._gview(u.p_bias))
                self._dgrad(grads, u, dt, u.out.c * u.src.c, dtp)
            elif u.kind == "upsample":
                if self._needs_grad[u.src.tid]:
                    cur = grads.get(u.src.tid)
                    if cur is None:
                        cur = torch.empty((N, u.src.h, u.src.w, u.src.c), device=self.device, dtype=torch.float32)
                        ops.upsample2x_bwd(dout, u.out.c, 0, cur, accumulate=False)
                        grads[u.src.tid] = cur
                    else:
                        ops.upsample2x_bwd(dout, u.out.c, 0, cur, accumulate=True)
            elif u.kind == "concat":
                off = 0
                for s in u.srcs:
                    if self._needs_grad[s.tid]:
                        cur = grads.get(s.tid)
                        if (cur is None and self._concat_grad_slices and self._n_consumers.get(s.tid, 0) == 1
                                and s.producer is not None and self._takes_grad_slice(s.producer)
                                and s.c % 8 == 0 and off % 4 == 0 and u.out.c % 4 == 0):
                            grads[s.tid] = ops.ChannelSlice(dout, u.out.c, off, s.c)   # read in place by bn_act_bwd
                            off += s.c
                            continue
                        if cur is None:
                            cur = torch.empty((N, s.h, s.w, s.c), device=self.device, dtype=torch.float32)
                            ops.copy_channels_out(dout, u.out.c, off, cur, s.c, accumulate=False)
                            grads[s.tid] = cur
                        else:
                            ops.copy_channels_out(dout, u.out.c, off, cur, s.c, accumulate=True)
                    off += s.c
            elif u.kind == "maxpool":
                if self._needs_grad[u.src.tid]:
                    cur = grads.get(u.src.tid)
                    if (u.k == 2 and u.stride == 2 and u.pad_t == 0 and u.pad_l == 0 and u.src.h == 2 * u.out.h
                            and u.src.w == 2 * u.out.w and u.out.c % 4 == 0 and torch.is_tensor(dout) and dout.is_contiguous()
                            and (cur is None or torch.is_tensor(cur))):
                        # the windows tile the input: every input position is written / added to once, no zero fill, no atomics
                        if cur is None:
                            cur = torch.empty((N, u.src.h, u.src.w, u.src.c), device=self.device, dtype=torch.float32)
                            grads[u.src.tid] = cur
                            ops.maxpool2x2_bwd(dout, N, u.out.h, u.out.w, u.out.c, u.argmax, cur, False)
                        else:
                            ops.maxpool2x2_bwd(dout, N, u.out.h, u.out.w, u.out.c, u.argmax, cur, True)
                        continue
                    if cur is None:
                        cur = torch.empty((N, u.src.h, u.src.w, u.src.c), device=self.device, dtype=torch.float32)
                        ops.zero_bytes(cur)
                        grads[u.src.tid] = cur
                    if u.stride == 1 and (u.out.h, u.out.w) == (u.src.h, u.src.w):
                        ops.maxpool_bwd_same(dout, N, u.out.h, u.out.w, u.out.c, u.out.c, 0, u.argmax, u.k, u.pad_t, u.pad_l, cur)
                    else:
                        ops.maxpool_bwd(dout, N, u.out.h, u.out.w, u.out.c, u.out.c, 0, u.argmax, cur)
            elif u.kind == "space_to_depth":
                if self._needs_grad[u.src.tid]:
                    cur = grads.get(u.src.tid)
                    if cur is None:
                        cur = torch.empty((N, u.src.h, u.src.w, u.src.c), device=self.device, dtype=torch.float32)
                        ops.space_to_depth2_bwd(dout, u.out.c, 0, cur, accumulate=False)
                        grads[u.src.tid] = cur
                    else:
                        ops.space_to_depth2_bwd(dout, u.out.c, 0, cur, accumulate=True)
            if self.grad_ready_hook is not None and u.kind in ("conv", "head"):
                tape.host_call(lambda u=u: self.grad_ready_hook(u))
        self._join_wgrad()

    # The filter gradient of a layer is independent of everything the backward pass does next (its data
    # gradient, the BatchNorm backward of the layer below, ...): it is enqueued on a second stream so that it
    # shares the chip with them -- the conv kernels alone leave workgroup slots idle in their last round (344
    # or 676 tiles on 512 slots) and the BatchNorm kernels leave the matrix pipe idle. Hazards: the dy planes
    # scratch is double-buffered and re-used only after the filter gradient that read it has finished (event);
    # fp32 tensors read on the second stream are marked with record_stream; the main stream joins the second
    # one at the end of backward. A grad_ready_hook (data parallel gradient buckets) must order its work after
    # BOTH streams (dp.GradReducer.extra_streams).
    def _next_dyp_buffer(self, u=None):
        if u is not None and u.name in self._dyp_own:
            self._dyp_cur_own = True
            return self._dyp_own[u.name]
        self._dyp_cur_own = False
        self._dyp_idx ^= 1
        ev = self._dyp_events[self._dyp_idx]
        if ev is not None:
            tape.wait_event(torch.cuda.current_stream(), ev)
            self._dyp_events[self._dyp_idx] = None
        if self._dyplanes2[self._dyp_idx] is None and self._dyp_shared_bytes:
            self._dyplanes2[self._dyp_idx] = torch.empty(self._dyp_shared_bytes, device=self.device, dtype=torch.uint8)
        return self._dyplanes2[self._dyp_idx]

    @contextlib.contextmanager
    def _beside_backward(self, *tensors):
        if not self._overlap_wgrad:
            yield
            return
        if self._wgrad_stream is None:
            self._wgrad_stream = ops.concurrent_stream("wgrad", device=self.device)
        side = self._wgrad_stream
        tape.wait_stream(side, torch.cuda.current_stream())
        for t in tensors:
            if t is not None:
                t.record_stream(side)
        with torch.cuda.stream(side):
            yield
            # (the event guards the REUSE of a shared dy planes buffer: a layer with planes of its own needs none)
            ev = None if getattr(self, "_dyp_cur_own", False) else tape.record_event(side)
        if ev is not None:
            self._dyp_events[self._dyp_idx] = ev
        self._wgrad_pending = True

    def _join_wgrad(self):
        if self._wgrad_pending:
            tape.wait_stream(torch.cuda.current_stream(), self._wgrad_stream)
            self._wgrad_pending = False

    def _dyp(self, u, dy):
        """planes of this layer's dy (one scratch, consumed by the filter and data gradients right away)"""
        if not (u.planes_wgrad or u.planes_dgrad):
            return None
        cout = u.cout if u.kind == "conv" else u.out.c
        return ops.split_planes(dy, self.batch * u.out.h * u.out.w, cout, out=self._next_dyp_buffer(u))

    def _bnred_of(self, w, cur):
        """the fused reduction this writer's data gradient makes (None: none). cur = the gradient it adds into, if any: a
        ChannelSlice or a tensor that is not dense NHWC of the source never takes the fused form"""
        p = w.bnred_for
        if p is None or p.bnred is None or (cur is not None and not torch.is_tensor(cur)):
            return None
        p.bnred_done = True
        return p.bnred

    def _take_fused(self, u, dout):
        """the BnReduce that already holds the reduction of `dout` for unit u's BatchNormalization backward, or None"""
        b = getattr(u, "bnred", None)
        if b is None or not u.bnred_done:
            return None
        u.bnred_done = False
        if not torch.is_tensor(dout):
            raise YoloHipError(f"{u.name}: fused reduction made for a gradient that arrives as a channel slice")
        return b

    def _dgrad(self, grads, u, dy, wsize, dyp=None):
        if not self._needs_grad[u.src.tid]:
            return
        cur = grads.get(u.src.tid)
        if u.planes_dgrad:
            wTp = self._wplanes[u.wTp_off:u.wTp_off + u.wTp_bytes]
            b = self._bnred_of(u, cur)
            if cur is None:
                grads[u.src.tid] = ops.conv2d_dgrad_planes(u.desc, dyp, wTp, bnred=b)
            else:
                ops.conv2d_dgrad_planes(u.desc, dyp, wTp, dx=cur, accumulate=True, bnred=b)
            return
        wT = self._wT[u.wT_off:u.wT_off + wsize]
        if cur is None:
            grads[u.src.tid] = ops.conv2d_dgrad(u.desc, dy, wT)
        else:
            ops.conv2d_dgrad(u.desc, dy, wT, dx=cur, accumulate=True)

    # ---- weights ------------------------------------------------------------------------
    def named_weights(self):
        """name -> numpy array in KERAS layout (kernel HWIO), for .npz interchange."""
        out = {}
        for store in (self.params, self.state):
            for name in store.order:
                s = store.specs[name]
                a = store.view(name).detach().cpu().numpy().reshape(s.shape)
                if name.endswith("/kernel"):
                    a = np.transpose(a, (1, 2, 3, 0))  # KRSC -> HWIO
                out[name] = a.copy()
        return out

    def load_named_weights(self, weights, strict=True):
        self.before_param_write()
        for store in (self.params, self.state):
            for name in store.order:
                if name not in weights:
                    if strict:
                        raise KeyError(f"missing weight {name}")
                    continue
                s = store.specs[name]
                a = np.asarray(weights[name], dtype=np.float32)
                if name.endswith("/kernel"):
                    a = np.transpose(a, (3, 0, 1, 2))  # HWIO -> KRSC
                if tuple(a.shape) != s.shape:
                    raise ValueError(f"{name}: shape {a.shape} != {s.shape}")
                store.view(name).copy_(torch.from_numpy(np.ascontiguousarray(a)).reshape(-1))
        self.mark_params_changed()
