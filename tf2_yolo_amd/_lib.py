"""ctypes binding of libyolo_hip.so (C-ABI declared in include/yolo_hip.h).

The product path has NO CPU fallback: if the HIP library is missing or a call fails,
`YoloHipError` is raised. (The CPU restatement under /oracle is test infrastructure and
is never imported from here.)
"""
import ctypes
import os
from ctypes import (POINTER, Structure, c_char_p, c_double, c_float, c_int, c_longlong,
                    c_size_t, c_void_p)

from . import tape as _tape

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libyolo_hip.so")

ACT_LINEAR, ACT_LEAKY, ACT_MISH = 0, 1, 2
NMS_HARD, NMS_SOFT, NMS_DIOU = 1, 2, 3


class YoloHipError(RuntimeError):
    pass


class ConvDesc(Structure):
    _fields_ = [(n, c_int) for n in
                ("N", "H", "W", "Cin", "Cout", "kh", "kw", "Ho", "Wo", "sh", "sw", "pad_t", "pad_l")]


class LossCfg(Structure):
    _fields_ = [("version", c_int), ("N", c_int), ("gh", c_int), ("gw", c_int), ("A", c_int),
                ("C", c_int), ("anchors", c_float * 32), ("use_anchors", c_int),
                ("binary_weight", c_float), ("loss_weight", c_float * 4),
                ("ignore_thresh", c_float), ("use_focal_loss", c_int), ("focal_gamma", c_float),
                ("use_scale", c_int), ("wh_reg_weight", c_float), ("truth_thresh", c_float),
                ("label_smooth", c_float)]


_P = c_void_p
_LL = c_longlong
# name -> (restype, argtypes); must list EVERY symbol declared in include/yolo_hip.h
SIGNATURES = {
    "yolo_last_error": (c_char_p, []),
    "yolo_abi_version": (c_int, []),
    "yolo_device_available": (c_int, []),
    "yolo_set_option": (c_int, [c_int, c_int]),
    "yolo_mfma_probe": (c_int, [_P, _P, c_int, c_int, POINTER(c_double), _P]),
    "yolo_set_debug_buffer": (c_int, [_P, c_size_t]),
    "yolo_encode_labels": (c_int, [_P, _P, _P, c_int, c_double, c_double, c_int, c_int, c_int, _P, _P, _P]),
    "yolo_down2xlabel": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "yolo_conv_workspace_bytes": (c_size_t, []),
    "yolo_set_conv_workspace": (c_int, [_P, c_size_t, _P]),
    "yolo_conv2d_fwd": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P]),
    "yolo_conv2d_dgrad": (c_int, [POINTER(ConvDesc), _P, _P, _P, c_int, _P]),
    "yolo_conv2d_wgrad": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P]),
    "yolo_conv2d_wgrad_bias": (c_int, [_P, _LL, c_int, _P, _P]),
    "yolo_filter_transpose": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "yolo_planes_bytes": (c_size_t, [_LL, c_int]),
    "yolo_split_planes": (c_int, [_P, _LL, c_int, _P, _P]),
    "yolo_split_planes_padded": (c_int, [_P, _LL, c_int, c_int, _P, _P]),
    "yolo_split_planes_batch": (c_int, [_P, c_int, _LL, _P]),
    "yolo_filter_transpose_batch": (c_int, [_P, c_int, _LL, _P]),
    "yolo_conv2d_fwd_planes": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P]),
    "yolo_stem_bwd_scratch_bytes": (c_size_t, []),
    "yolo_stem_bn_bwd_wgrad": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, c_size_t, _P]),
    "yolo_conv2d_fwd_planes_epi": (c_int, [POINTER(ConvDesc), _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P]),
    "yolo_split_planes_absmax": (c_int, [_P, _LL, c_int, _P, _P, _P, _P, _P]),
    "yolo_conv_pred_bound": (c_int, [_P, c_int, c_int, _P, _P, _P, _P, _P]),
    "yolo_conv2d_fwd_infer_unit": (c_int, [POINTER(ConvDesc), _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P, c_int, _P, c_int,
                                           _P, _P, _P, POINTER(c_int), _P]),
    "yolo_fold_bound": (c_int, [_P, c_int, _P, _P]),
    "yolo_absmax_words": (c_int, [_P, _LL, _P, POINTER(c_int), _P]),
    "yolo_stem_filter_prep": (c_int, [_P, _P, _P, _P]),
    "yolo_stem_fwd_infer_unit": (c_int, [POINTER(ConvDesc), _P, _P, c_int, _P, _P, _P, _P, c_int, _P, _P, _P, POINTER(c_int), _P]),
    "yolo_split_planes_concat": (c_int, [POINTER(c_void_p), POINTER(c_int), POINTER(c_void_p), c_int, _LL, _P, _P, _P, _P]),
    "yolo_split_planes_concat_ex": (c_int, [POINTER(c_void_p), POINTER(c_int), POINTER(c_void_p), POINTER(c_int), POINTER(c_int),
                                            c_int, c_int, c_int, _LL, _P, _P, _P, _P]),
    "yolo_conv2d_fwd_head_unit": (c_int, [POINTER(ConvDesc), _P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P]),
    "yolo_conv2d_fwd_absmax": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P, _P, _P, _P]),
    "yolo_conv2d_dgrad_planes": (c_int, [POINTER(ConvDesc), _P, _P, _P, c_int, _P]),
    "yolo_conv2d_wgrad_planes": (c_int, [POINTER(ConvDesc), _P, _P, _P, _P]),
    "yolo_wgrad_workspace_bytes": (c_size_t, []),
    "yolo_set_wgrad_workspace": (c_int, [_P, c_size_t]),
    "yolo_bn_stats": (c_int, [_P, _LL, c_int, _P, _P]),
    "yolo_bn_finalize": (c_int, [_P, _LL, c_int, _P, _P, c_float, c_float, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "yolo_bn_fold_inference": (c_int, [c_int, _P, _P, _P, _P, c_float, _P, _P, _P]),
    "yolo_bn_act_fwd": (c_int, [_P, _LL, c_int, _P, _P, c_int, _P, _P, _P]),
    "yolo_bn_act_bwd_reduce": (c_int, [_P, _P, _LL, c_int, _P, _P, _P, _P, c_int, _P, _P]),
    "yolo_bn_act_bwd_apply": (c_int, [_P, _P, _LL, c_int, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P]),
    "yolo_bn_act_fwd_planes": (c_int, [_P, _LL, c_int, _P, _P, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "yolo_bn_act_fwd_res_planes": (c_int, [_P, _LL, c_int, _P, _P, c_int, _P, _P, _P, _P, _P, _P]),
    "yolo_bn_act_bwd_apply_planes": (c_int, [_P, _P, _LL, c_int, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P,
                                             _P]),
    "yolo_bn_infer_bound": (c_int, [c_int, _P, _P, _P, _P, _P]),
    "yolo_bn_finalize_bound": (c_int, [_P, _LL, c_int, _P, _P, c_float, c_float, c_int, _P, _P, _P, _P, _P, _P, _P,
                                       _P, _P]),
    "yolo_bn_finalize_offset": (c_int, [_P, _LL, c_int, _P, _P, c_float, c_float, c_int, _P, _P, _P, _P, _P, _P, _P,
                                        _P, _P, _P]),
    "yolo_bn_act_bwd_reduce_bound": (c_int, [_P, _P, _LL, c_int, _P, _P, _P, _P, c_int, _P, _P, _P]),
    "yolo_bn_act_bwd_reduce_bound_ld": (c_int, [_P, _P, _LL, _LL, c_int, _P, _P, _P, _P, c_int, _P, _P, _P]),
    "yolo_bn_act_bwd_reduce_fold_ld": (c_int, [_P, _P, _LL, _LL, c_int, _P, _P, _P, _P, c_int, _P, _P, _P, _P]),
    "yolo_bn_act_bwd_apply_planes_ld": (c_int, [_P, _P, _LL, _LL, c_int, _P, _P, _P, _P, _P, c_int, _P, _P, _P, _P, _P, _P,
                                                _P]),
    "yolo_bnred_slots_cap": (c_int, [POINTER(ConvDesc)]),
    "yolo_conv2d_dgrad_planes_bnred": (c_int, [POINTER(ConvDesc), _P, _P, _P, c_int, _P, _P, _P, _P, _P, c_int, _P, c_int, _P,
                                               POINTER(c_int), _P]),
    "yolo_bn_act_bwd_sum_partials": (c_int, [_P, c_int, _LL, c_int, _P, _P, _P, _P]),
    "yolo_allreduce_bucket": (c_int, [_P, _P, _LL, _P]),
    "yolo_act_fwd": (c_int, [_P, _LL, c_int, _P, _P]),
    "yolo_act_bwd": (c_int, [_P, _P, _LL, c_int, _P, _P]),
    "yolo_copy_channels_in": (c_int, [_P, _LL, c_int, _P, c_int, c_int, _P]),
    "yolo_copy_channels_out": (c_int, [_P, _LL, c_int, c_int, _P, c_int, c_int, _P]),
    "yolo_upsample2x_fwd": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P]),
    "yolo_upsample2x_bwd": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "yolo_axpy": (c_int, [_P, _P, _LL, _P]),
    "yolo_maxpool_fwd": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int,
                                 _P, c_int, c_int, _P, _P]),
    "yolo_maxpool_bwd": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, _P, _P]),
    "yolo_maxpool2x2_bwd": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, c_int, _P]),
    "yolo_bn_act_maxpool2x2_fwd": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P, _P, _P, _P, _P]),
    "yolo_maxpool_bwd_same": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, c_int, c_int, _P, _P]),
    "yolo_space_to_depth2_fwd": (c_int, [_P, c_int, c_int, c_int, c_int, _P, c_int, c_int, _P]),
    "yolo_space_to_depth2_bwd": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P]),
    "yolo_head_act_fwd": (c_int, [_P, _LL, c_int, c_int, c_int, _P, _P, _P]),
    "yolo_head_act_bwd": (c_int, [_P, _P, _LL, c_int, c_int, c_int, _P, _P, _P, _P]),
    "yolo_loss_workspace_bytes": (c_size_t, [POINTER(LossCfg)]),
    "yolo_loss_fwd_bwd": (c_int, [POINTER(LossCfg), _P, _P, _P, _P, c_float, _P, c_size_t, _P]),
    "yolo_metrics": (c_int, [POINTER(LossCfg), _P, _P, c_float, _P, _P]),
    "yolo_adam_step": (c_int, [_P, _P, _P, _P, _LL, c_float, c_float, c_float, c_float, c_int, c_float, c_int, _P]),
    "yolo_adam_lr_t": (c_float, [c_float, c_float, c_float, c_int]),
    "yolo_adam_step_dev": (c_int, [_P, _P, _P, _P, _LL, _P, c_int, _P]),
    "yolo_sgd_step": (c_int, [_P, _P, _LL, c_float, c_float, c_int, _P]),
    "yolo_fill": (c_int, [_P, _LL, c_float, _P]),
    "yolo_decode_level": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_float, _P, c_int, _P, _P, c_size_t, _P]),
    "yolo_decode_level_f64": (c_int, [_P, c_int, c_int, c_int, c_int, c_int, c_double, _P, c_int, _P, _P,
                                      c_size_t, _P]),
    "yolo_decode_workspace_bytes": (c_size_t, [c_int, c_int, c_int, c_int]),
    "yolo_nms_workspace_bytes": (c_size_t, [c_int, c_int]),
    "yolo_cal_iou": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, POINTER(c_longlong), POINTER(c_longlong),
                             POINTER(c_longlong), c_double, c_double, _P]),
    "yolo_match_detections": (c_int, [_P, c_int, _P, c_int, c_int, c_double, _P, _P, _P, _P, _P, _P]),
    "yolo_rank_desc": (c_int, [_P, _P, c_int, _P, _P]),
    "yolo_pr_curve_workspace_bytes": (c_size_t, [c_int, c_int]),
    "yolo_pr_curve": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P, c_size_t, _P, _P, _P]),
    "yolo_nms": (c_int, [_P, c_int, c_int, c_int, c_double, c_double, c_double, _P, _P, c_size_t, _P]),
    "yolo_nms_select": (c_int, [_P, c_int, c_int, c_int, c_double, c_double, c_double, _P, _P, _P, _P, c_size_t, _P]),
}

_lib = None

# functions that only answer a question on the host: never part of a recorded step (tape.py)
_QUERIES = {"yolo_last_error", "yolo_abi_version", "yolo_bnred_slots_cap", "yolo_device_available", "yolo_conv_workspace_bytes", "yolo_planes_bytes",
            "yolo_stem_bwd_scratch_bytes", "yolo_loss_workspace_bytes", "yolo_decode_workspace_bytes",
            "yolo_nms_workspace_bytes", "yolo_pr_curve_workspace_bytes", "yolo_wgrad_workspace_bytes", "yolo_adam_lr_t",
            "yolo_set_option", "yolo_set_debug_buffer", "yolo_set_conv_workspace", "yolo_set_wgrad_workspace"}


class _Recorded:
    """a library function that, while a launch tape records (tape.ACTIVE), remembers (function, arguments) of every call"""
    __slots__ = ("f", "name")

    def __init__(self, f, name):
        self.f, self.name = f, name

    def __call__(self, *a):
        rc = self.f(*a)
        if _tape.ACTIVE is not None:
            _tape.ACTIVE.c(self.f, a, self.name)
        return rc


class _Lib:
    """attribute access like the ctypes.CDLL it wraps; enqueueing functions are recordable"""

    def __init__(self, cdll):
        self._cdll = cdll

    def __getattr__(self, name):
        f = getattr(self._cdll, name)
        if name not in _QUERIES and name in SIGNATURES:
            f = _Recorded(f, name)
        setattr(self, name, f)
        return f


def load():
    """Load libyolo_hip.so and declare every prototype. Raises YoloHipError if absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise YoloHipError(
            f"{LIB_PATH} not found: build it with `make` (or __graft_entry__.build()); "
            "there is no CPU fallback for the product path")
    # torch bundles its own HIP runtime under the same SONAME (libamdhip64.so.7). It must be mapped
    # BEFORE this library so that both resolve to ONE runtime (one device context, shared streams);
    # loading libyolo_hip.so first would pull in /opt/rocm's copy and give torch a second runtime.
    import torch  # noqa: F401
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise YoloHipError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise YoloHipError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = _Lib(lib)
    return _lib


def check(rc, what=""):
    if rc != 0:
        msg = load().yolo_last_error()
        raise YoloHipError(f"{what} failed (status {rc}): {msg.decode() if msg else ''}")
