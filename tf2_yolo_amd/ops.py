"""Host-side operator layer: torch tensors in, C-ABI calls out.

torch is plumbing only (device memory, streams). Every function enqueues hand-written HIP
kernels from libyolo_hip.so on torch's current stream and raises `YoloHipError` on failure;
nothing here computes on the CPU or through torch operators.
"""
import ctypes
from ctypes import byref, c_void_p

import torch

from . import _lib
from . import tape as _tape
from ._lib import ConvDesc, LossCfg, YoloHipError, check

BN_EPS = 1e-3       # Keras BatchNormalization default
BN_MOMENTUM = 0.99
BN_STAT_SLOTS = 64  # YOLO_BN_STAT_SLOTS in include/yolo_hip.h
BN_RED_SLOTS = 512
SPLIT_BATCH_UNITS = 4   # include/yolo_hip.h YOLO_SPLIT_BATCH_UNITS


class KernelTimer:
    """Brackets the MFMA conv launches with HIP events on the launch stream (bench.py roofline).
    Aggregates per kernel variant: launches, device time, algorithmic FLOPs (2*M*N*K, no padding,
    im2col or recompute credit). Enabled by assigning an instance to `ops.TIMER`."""

    def __init__(self):
        self.pending = []

    def bracket(self, name, flops, launches, fn, layer=None):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        fn()
        e.record()
        self.pending.append((name, flops, launches, s, e, layer))

    def summary(self, by_layer=False):
        """per kernel variant (default) or per (kernel variant, layer) -- layer = (pass, H, W, Cin, Cout, k, stride, N)"""
        torch.cuda.synchronize()
        agg = {}
        for name, flops, launches, s, e, layer in self.pending:
            a = agg.setdefault((name, layer) if by_layer else name, {"launches": 0, "ms": 0.0, "flops": 0.0})
            a["launches"] += launches
            a["ms"] += s.elapsed_time(e)
            a["flops"] += flops
        self.pending = []
        return agg


TIMER = None


def _conv_flops(d):
    return 2.0 * d.N * d.Ho * d.Wo * d.Cout * d.kh * d.kw * d.Cin


def _layer_key(d, which):
    return (which, d.H, d.W, d.Cin, d.Cout, d.kh, d.sh, d.N)


import os as _os
CONV_MODE = "fp32" if _os.environ.get("YOLO_CONV_MODE", "split")[:1] in ("f", "0") else "split"
# engine-level choice: keep conv operands as pre-split scaled fp16 planes (h + l) and use the LDS-DMA kernels
# (conv_planes.hip, conv_win.hip, conv_wgrad_planes.hip)
USE_PLANES = CONV_MODE == "split" and _os.environ.get("YOLO_CONV_PLANES", "1") != "0"


def planes_fwd_ok(cin, cout):
    return USE_PLANES and cin % 16 == 0 and cout >= 32


def planes_dgrad_ok(cin, cout):
    return USE_PLANES and cout % 16 == 0 and cin >= 32


def planes_wgrad_ok(cin, cout, taps, stride=1):
    # (stride is not a criterion: stride-2 filter gradients run on the planes kernel too)
    return USE_PLANES and cin % 16 == 0 and cout % 16 == 0 and cout >= 32 and taps * cin >= 64


_WGRAD_WIDE = int(_os.environ.get("YOLO_WGRAD_WIDE", "2"))


WGRAD_WIN = int(_os.environ.get("YOLO_WGRAD_WIN", "1"))


def _wgrad_win_covers(d):
    """mirrors wgrad_win_supported() in csrc/conv_wgrad_win.hip: 3x3 stride-1 'same', 128-filter x 32-channel tiles,
    at least 64 and fewer than 2^31 - 4096 pixels (WGRAD_WIN follows set_option / reset_options)"""
    m = d.N * d.Ho * d.Wo
    return (WGRAD_WIN != 0 and d.kh == 3 and d.kw == 3 and d.sh == 1 and d.sw == 1 and d.pad_t == 1 and d.pad_l == 1
            and d.H == d.Ho and d.W == d.Wo and d.Cout % 128 == 0 and d.Cin % 32 == 0 and d.W >= 4
            and 2 * d.W + 81 <= 512 and 32 // d.W + 2 <= d.H and 64 <= m < (1 << 31) - 4096)


def _wgrad_planes_variant(cout, cols, taps=1, pixels=1 << 40, d=None):
    """mirrors launch_wgrad_planes() in csrc/conv_wgrad_planes.hip (names used by bench.py's roofline report)"""
    if d is not None and _wgrad_win_covers(d):
        # conv_wgrad_win.hip: 128 filters x (9 taps x 32 channels) per workgroup, x streamed once through an LDS ring
        return "wgrad_win_planes_kernel<128,288%s>" % (",mfma16" if WGRAD_WIN == 3 else "")
    if cout > 64 and cols >= 256 and (_WGRAD_WIDE == 1 or (_WGRAD_WIDE == 2 and taps > 1 and pixels <= 8192)):
        return "wgrad_planes_kernel<128,256,2,2>"
    big = "4,2" if _os.environ.get("YOLO_WGRAD_WAVES") == "8" else "2,2"
    return "wgrad_planes_kernel<%d,%d,%s>" % (64 if cout <= 64 else 128, 64 if cols <= 64 else 128,
                                             "2,2" if (cout <= 64 and cols <= 64) else "2,4" if cout <= 64
                                             else "4,2" if cols <= 64 else big)


_PLANES_WAVES = int(_os.environ.get("YOLO_PLANES_WAVES", "0") or 0)   # 0: the library's automatic choice


CONV_WIN = int(_os.environ.get("YOLO_CONV_WIN", "1"))
CONV_PATCH = int(_os.environ.get("YOLO_CONV_PATCH", "1"))


def _planes_variant(cout, win=None, k1=False):
    """mirrors launch_gather_planes() in csrc/conv_planes.hip and launch_conv_win() in csrc/conv_win.hip;
    win = (W of the source, M rows) of a 3x3 stride-1 layer, None otherwise; k1 = 1x1 stride-1 layer (8-wave tile)"""
    if win is not None and CONV_WIN not in (0, 2, 4) and CONV_PATCH and cout >= 64:
        # launch_conv_patch() in csrc/conv_win.hip: rows longer than 64 pixels or Cout <= 64, at least 256 tiles
        ws, m, hs, n = win
        narrow = cout <= 64
        tiles = (n * ((hs + 15) // 16) * ((ws + 15) // 16) if narrow
                 else n * ((hs + 7) // 8) * ((ws + 15) // 16) * ((cout + 127) // 128))
        if CONV_PATCH == 2 or ((narrow or ws > 64) and tiles >= 256):
            return "conv_patch_planes_kernel<%s>" % ("256,64" if narrow else "128,128")
    if win is not None and CONV_WIN and cout >= 128:
        ws, m = win[:2]
        wgm = CONV_WIN if CONV_WIN in (2, 4) else (0 if ws > 64 else 2)
        if wgm:
            return "conv_win_planes_kernel<%d,128>" % (64 * wgm)
    if cout <= 32:
        return "gather_conv_planes_kernel<128,32,4,1>"
    if cout <= 64:
        return "gather_conv_planes_kernel<128,64,4,2>"
    waves8 = _PLANES_WAVES == 8 or (_PLANES_WAVES == 0 and k1)
    return "gather_conv_planes_kernel<128,128,%s>" % ("4,2" if waves8 else "2,2")


def _win_key(d, hs, ws, cs):
    """(W, M) when the window kernel covers the layer: 3x3, stride 1, 'same', 16-channel blocks"""
    if d.kh == 3 and d.kw == 3 and d.sh == 1 and d.sw == 1 and d.pad_t == 1 and d.pad_l == 1 and cs % 16 == 0 \
            and d.H == d.Ho and d.W == d.Wo:
        return (ws, d.N * hs * ws, hs, d.N)
    return None


def _gather_variant(cout, flat, m=None):
    """mirrors dispatch_gather() in csrc/conv.hip (names used by bench.py's roofline report)"""
    bn = 32 if cout <= 32 else (64 if cout <= 64 else 128)
    bm = 128
    if bn == 128 and not flat and m is not None and ((m + 127) // 128) * ((cout + 127) // 128) <= 512:
        bm = 64
    if CONV_MODE == "split" and not flat and cout > 32:
        waves = {(128, 64): "2,2", (64, 128): "1,4", (128, 128): "4,2"}[(bm, bn)]
        return f"gather_conv_split_kernel<{bm},{bn},{waves}>"
    return f"gather_conv_kernel<{bm},{bn}{',flat' if flat else ''}>"


OPT_CONV_WIN = 0   # include/yolo_hip.h YOLO_OPT_CONV_WIN
OPT_CONV_SK = 2
OPT_CONV_PATCH = 5   # include/yolo_hip.h YOLO_OPT_CONV_PATCH
OPT_WGRAD_WIN = 6    # 3x3 stride-1 filter gradient with the input window in LDS (conv_wgrad_win.hip): 0 off, 1 on
OPT_NMS_WALK = 7     # hard / DIoU NMS: 1 = the greedy walk kernel for every class (default 0: pair bit matrix + walk over the bits)
_CONV_WS = None


def ensure_conv_workspace():
    """scratch of the persistent (stream-K) window kernel: allocated once per process, registered with the library"""
    global _CONV_WS
    # The library's per-process state (this workspace, cached kernel attributes / occupancy, the stem's filter ring)
    # belongs to ONE device: one process drives one GPU (torchrun starts a process per GPU).
    # Launches that use the workspace (split-K / stream-K forward and data-gradient convolutions) must be ordered on
    # one stream; the executor's second stream only runs filter gradients and filter preparation, which do not.
    if _CONV_WS is not None and _CONV_WS.device.index != torch.cuda.current_device():
        raise YoloHipError(f"libyolo_hip.so was initialised on cuda:{_CONV_WS.device.index} and cannot also serve "
                           f"cuda:{torch.cuda.current_device()}: start one process per GPU")
    if _CONV_WS is None:
        lib = _lib.load()
        n = int(lib.yolo_conv_workspace_bytes())
        _CONV_WS = torch.empty(n, dtype=torch.uint8, device="cuda")
        check(lib.yolo_set_conv_workspace(_p(_CONV_WS), n, _stream()), "yolo_set_conv_workspace")
    return _CONV_WS


_WGRAD_WS = None


def ensure_wgrad_workspace():
    """scratch of the reproducible (atomics-free) filter / bias gradient reductions (csrc/conv_wgrad_planes.hip): allocated
    once per process, registered with the library. Launches that use it must be ordered on one stream (the executor runs
    every filter gradient on its second stream, or all of them on the main stream)."""
    global _WGRAD_WS
    if _WGRAD_WS is None:
        lib = _lib.load()
        n = int(lib.yolo_wgrad_workspace_bytes())
        _WGRAD_WS = torch.empty(n, dtype=torch.uint8, device="cuda")
        check(lib.yolo_set_wgrad_workspace(_p(_WGRAD_WS), n), "yolo_set_wgrad_workspace")
    return _WGRAD_WS


# ---- side streams of the process: created ONCE, EARLY, in a fixed order ----
# The ROCm runtime maps HIP streams onto a few hardware queues (GPU_MAX_HW_QUEUES; tf2_yolo_amd/__init__.py raises the default
# of 4 to 8) in creation order, wrapping around, and two streams on one queue execute one after the other. Measured (round 4,
# profiles/r04_b_dp_*): with RCCL initialised first, the filter-gradient stream -- created lazily in the first forward pass --
# landed on the compute stream's queue: 34.2 instead of 30.1 ms per step; as the fourth model of one process, YOLOv4-608's own
# filter-gradient stream did the same: 58.8 instead of 39.7 ms. So the process has ONE stream per role, shared by every model,
# and the roles are created together, filter gradients first: call create_side_streams() before anything else creates streams
# (bench.py does, before init_process_group; Network.__init__ does for single-process use). YOLO_STREAM_PROBE=1 additionally
# checks with two spin kernels that the new stream overlaps the current one and warns if not (diagnostic; the probe's extra
# streams shift later assignments, so it is off by default).
_SIDE_STREAMS = {}


def _overlaps(a, b, cycles=6000000):   # ~3 ms per spin kernel: far above the ~50 us of the cross-stream waits around them
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize()
    with torch.cuda.stream(a):
        e0.record(a)
        torch.cuda._sleep(cycles)
        e1.record(a)
    torch.cuda.synchronize()
    one = e0.elapsed_time(e1)
    with torch.cuda.stream(a):
        e0.record(a)
    b.wait_event(e0)
    with torch.cuda.stream(b):
        torch.cuda._sleep(cycles)
    with torch.cuda.stream(a):
        torch.cuda._sleep(cycles)
    a.wait_stream(b)
    e2.record(a)
    torch.cuda.synchronize()
    return e0.elapsed_time(e2) < 1.5 * one


def create_side_streams(device=None):
    """the filter-gradient and communication streams of `device` (default: the current device), in that order (idempotent;
    one pair per device of the process, created on that device explicitly)"""
    dev = torch.cuda.current_device() if device is None else torch.device(device).index
    if dev is None:
        dev = torch.cuda.current_device()
    for role in ("wgrad", "comm"):
        if (dev, role) not in _SIDE_STREAMS:
            st = torch.cuda.Stream(device=dev)
            if (_os.environ.get("YOLO_STREAM_PROBE") == "1" and dev == torch.cuda.current_device()
                    and not _overlaps(torch.cuda.current_stream(), st)):
                import warnings
                warnings.warn(f"tf2_yolo_amd: the {role} stream shares a hardware queue with the compute stream: the step "
                              "will run them one after the other (create_side_streams() earlier, or raise GPU_MAX_HW_QUEUES)")
            _SIDE_STREAMS[(dev, role)] = st
    return {role: _SIDE_STREAMS[(dev, role)] for role in ("wgrad", "comm")}


def concurrent_stream(role, beside=None, device=None):
    """the side stream of `role` ('wgrad', 'comm') on `device` (default: the current device)"""
    st = create_side_streams(device)[role]
    want = torch.cuda.current_device() if device is None else torch.device(device).index
    if want is not None and st.device.index != want:
        raise YoloHipError(f"side stream of role {role!r} lives on cuda:{st.device.index}, not on cuda:{want}")
    return st


def set_option(key, value):
    """yolo_set_option: run-time kernel-variant switches of the library (benchmarks, tests)"""
    global WGRAD_WIN
    check(_lib.load().yolo_set_option(int(key), int(value)), "yolo_set_option")
    if int(key) == OPT_WGRAD_WIN:     # (the kernel names the timer / roofline report use follow the option)
        WGRAD_WIN = int(value)


def reset_options():
    """every option back to its default (the YOLO_* environment variables)"""
    global WGRAD_WIN
    check(_lib.load().yolo_set_option(-1, 0), "yolo_set_option")
    WGRAD_WIN = int(_os.environ.get("YOLO_WGRAD_WIN", "1"))


def _stream():
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    if t is None:
        return c_void_p(0)
    if _tape.ACTIVE is not None:
        _tape.ACTIVE.keep.append(t)     # its address is about to enter a recorded argument tuple: it must outlive the tape
    return c_void_p(t.data_ptr())


def _chk_f32(*ts):
    for t in ts:
        if t is None:
            continue
        if not (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise YoloHipError("expected a contiguous float32 CUDA tensor, got "
                               f"{t.dtype} {t.device} contiguous={t.is_contiguous()}")


def same_pad(size, k, s):
    """Keras/TF 'same' padding: returns (out, pad_before); the smaller half goes before."""
    out = -(-size // s)
    total = max((out - 1) * s + k - size, 0)
    return out, total // 2


def conv_desc(x_shape, cout, kh, kw, stride, padding):
    """padding: 'same' | 'valid' | 'darknet_s2' (ZeroPadding2D(((1,0),(1,0))) + valid)."""
    n, h, w, cin = x_shape
    if padding == "same":
        ho, pt = same_pad(h, kh, stride)
        wo, pl = same_pad(w, kw, stride)
    elif padding == "valid":
        ho, wo, pt, pl = (h - kh) // stride + 1, (w - kw) // stride + 1, 0, 0
    elif padding == "darknet_s2":
        ho, wo, pt, pl = (h + 1 - kh) // stride + 1, (w + 1 - kw) // stride + 1, 1, 1
    else:
        raise ValueError(f"bad padding {padding}")
    return ConvDesc(n, h, w, cin, cout, kh, kw, ho, wo, stride, stride, pt, pl)


def conv2d_fwd(d, x, w, bias=None, out=None, stats=None, absmax=None):
    _chk_f32(x, w, bias)
    if out is None:
        out = torch.empty((d.N, d.Ho, d.Wo, d.Cout), device=x.device, dtype=torch.float32)
    if x.numel() != d.N * d.H * d.W * d.Cin or w.numel() != d.Cout * d.kh * d.kw * d.Cin:
        raise YoloHipError("conv2d_fwd: tensor sizes do not match the descriptor")
    if out.numel() != d.N * d.Ho * d.Wo * d.Cout:
        raise YoloHipError("conv2d_fwd: output size does not match the descriptor")
    if stats is not None and stats.numel() != BN_STAT_SLOTS * 2 * d.Cout:
        raise YoloHipError("conv2d_fwd: stats must hold BN_STAT_SLOTS x 2 x Cout doubles")
    def run():
        check(_lib.load().yolo_conv2d_fwd_absmax(byref(d), _p(x), _p(w), _p(bias), _p(out), _p(stats), _p(absmax),
                                                 _stream()), "yolo_conv2d_fwd")
    if TIMER is not None:
        # (csrc/stem.hip: stem_fwd_supported -- the direct Cin = 3 kernel takes the first layer at training sizes)
        stem = (CONV_MODE == "split" and _os.environ.get("YOLO_STEM_DIRECT", "1") != "0" and d.Cin == 3 and d.Cout == 32
                and d.kh == 3 and d.kw == 3 and d.sh == 1 and d.sw == 1 and d.Ho == d.H and d.Wo == d.W
                and d.N * d.H * d.W >= (1 << 20))
        name = "stem_conv3x3_kernel" if stem else _gather_variant(d.Cout, d.Cin % 32 != 0, d.N * d.Ho * d.Wo)
        TIMER.bracket(name, _conv_flops(d), 1, run)
    else:
        run()
    return out


def planes_bytes(rows, c):
    n = int(_lib.load().yolo_planes_bytes(int(rows), int(c)))
    if n == 0:
        raise YoloHipError(f"planes: unsupported shape rows={rows} C={c} (C must be a multiple of 16)")
    return n


def split_planes(x, rows, c, out=None):
    """fp32 [rows][c] -> two scaled fp16 planes h + l (22-23 significant bits; include/yolo_hip.h: yolo_split_planes);
    returns a uint8 buffer"""
    _chk_f32(x)
    if x.numel() != rows * c:
        raise YoloHipError("split_planes: tensor size does not match rows x C")
    n = planes_bytes(rows, c)
    if out is None:
        out = torch.empty(n, device=x.device, dtype=torch.uint8)
    elif out.numel() < n or out.dtype != torch.uint8:
        raise YoloHipError("split_planes: output buffer too small")
    check(_lib.load().yolo_split_planes(_p(x), int(rows), int(c), _p(out), _stream()), "yolo_split_planes")
    return out


def split_planes_padded(x, rows, c_src, out=None):
    """fp32 [rows][c_src], c_src % 16 != 0 -> planes of [rows][ceil16(c_src)], zero columns past c_src
    (include/yolo_hip.h: yolo_split_planes_padded)"""
    _chk_f32(x)
    c = (c_src + 15) // 16 * 16
    if x.numel() != rows * c_src:
        raise YoloHipError("split_planes_padded: tensor size does not match rows x C_src")
    n = planes_bytes(rows, c)
    if out is None:
        out = torch.empty(n, device=x.device, dtype=torch.uint8)
    elif out.numel() < n or out.dtype != torch.uint8:
        raise YoloHipError("split_planes_padded: output buffer too small")
    check(_lib.load().yolo_split_planes_padded(_p(x), int(rows), int(c_src), int(c), _p(out), _stream()),
          "yolo_split_planes_padded")
    return out


class BatchJobs:
    """Device job table for yolo_split_planes_batch / yolo_filter_transpose_batch (built once, re-run every step)."""

    def __init__(self, kind, device):
        self.kind, self.device = kind, device
        self.rows, self.blocks, self.table = [], 0, None

    def add_split(self, src, dst, rows, c, bound_from=None):
        """bound_from: an already split planes buffer (uint8 tensor, exact size) holding the same values in
        another order - its bound is reused instead of a max|x| pass over `src`."""
        nb = -(-((rows + 15) // 16 + 1) // SPLIT_BATCH_UNITS) * -(-c // 128)
        donor = 0 if bound_from is None else bound_from.data_ptr() + bound_from.numel() - 256
        self.rows.append([src.data_ptr(), dst.data_ptr(), rows, c, donor, self.blocks])
        self.blocks += nb

    def add_transpose(self, src, dst, cout, taps, cin):
        nb = -(-cin // 32) * -(-cout // 32) * taps
        self.rows.append([src.data_ptr(), dst.data_ptr(), cout, taps, cin, self.blocks])
        self.blocks += nb

    def run(self):
        if not self.rows:
            return
        if self.table is None:
            self.table = torch.tensor(self.rows, dtype=torch.int64).to(self.device)
        fn = _lib.load().yolo_split_planes_batch if self.kind == "split" else _lib.load().yolo_filter_transpose_batch
        check(fn(_p(self.table), len(self.rows), self.blocks, _stream()), "yolo_%s_batch" % self.kind)


def conv2d_fwd_planes(d, xp, wp, bias=None, out=None, stats=None, absmax=None):
    _chk_f32(bias)
    if out is None:
        out = torch.empty((d.N, d.Ho, d.Wo, d.Cout), device=xp.device, dtype=torch.float32)
    if xp.numel() < planes_bytes(d.N * d.H * d.W, d.Cin) or wp.numel() < planes_bytes(d.Cout, d.kh * d.kw * d.Cin):
        raise YoloHipError("conv2d_fwd_planes: planes buffers do not match the descriptor")
    if out.numel() != d.N * d.Ho * d.Wo * d.Cout:
        raise YoloHipError("conv2d_fwd_planes: output size does not match the descriptor")
    if stats is not None and stats.numel() != BN_STAT_SLOTS * 2 * d.Cout:
        raise YoloHipError("conv2d_fwd_planes: stats must hold BN_STAT_SLOTS x 2 x Cout doubles")
    def run():
        check(_lib.load().yolo_conv2d_fwd_planes(byref(d), _p(xp), _p(wp), _p(bias), _p(out), _p(stats), _p(absmax),
                                                 _stream()), "yolo_conv2d_fwd_planes")
    if TIMER is not None:
        TIMER.bracket(_planes_variant(d.Cout, _win_key(d, d.H, d.W, d.Cin), d.kh * d.kw == 1), _conv_flops(d), 1, run,
                      _layer_key(d, "fwd"))
    else:
        run()
    return out


EPI_NONE, EPI_AFFINE, EPI_AFFINE_LEAKY, EPI_AFFINE_MISH = 0, 1, 2, 3   # include/yolo_hip.h YOLO_EPI_*


def conv2d_fwd_planes_epi(d, xp, wp, bias, epilogue, scale, shift, residual=None, out=None, absmax=None):
    """inference: out = act(scale * (conv + bias) + shift) (+ residual) in ONE kernel (yolo_conv2d_fwd_planes_epi);
    absmax: per-channel max|out| before the residual (bit patterns, zeroed by the caller)"""
    _chk_f32(bias, scale, shift, residual)
    if out is None:
        out = torch.empty((d.N, d.Ho, d.Wo, d.Cout), device=xp.device, dtype=torch.float32)
    if xp.numel() < planes_bytes(d.N * d.H * d.W, d.Cin) or wp.numel() < planes_bytes(d.Cout, d.kh * d.kw * d.Cin):
        raise YoloHipError("conv2d_fwd_planes_epi: planes buffers do not match the descriptor")
    if out.numel() != d.N * d.Ho * d.Wo * d.Cout or (residual is not None and residual.numel() != out.numel()):
        raise YoloHipError("conv2d_fwd_planes_epi: output / residual size does not match the descriptor")
    def run():
        check(_lib.load().yolo_conv2d_fwd_planes_epi(byref(d), _p(xp), _p(wp), _p(bias), int(epilogue), _p(scale), _p(shift),
                                                     _p(residual), _p(out), _p(absmax), _stream()),
              "yolo_conv2d_fwd_planes_epi")
    if TIMER is not None:
        TIMER.bracket(_planes_variant(d.Cout, _win_key(d, d.H, d.W, d.Cin), d.kh * d.kw == 1), _conv_flops(d), 1, run,
                      _layer_key(d, "fwd"))
    else:
        run()
    return out


def conv_pred_bound(w, cout, kdim, scale, shift, bias, pred2):
    """pred2 (2 device floats, zeroed by the caller) <- {K, D} of a conv-BN unit: |act(scale (conv(x) + bias) + shift)| <=
    K max|x| + D (include/yolo_hip.h: yolo_conv_pred_bound)"""
    _chk_f32(w, scale, shift, bias, pred2)
    if w.numel() != cout * kdim or pred2.numel() != 2:
        raise YoloHipError("conv_pred_bound: w must hold Cout x kdim floats, pred2 two")
    check(_lib.load().yolo_conv_pred_bound(_p(w), int(cout), int(kdim), _p(scale), _p(shift), _p(bias), _p(pred2), _stream()),
          "yolo_conv_pred_bound")


INFER_BOUND_WORDS = 4096   # YOLO_INFER_BOUND_WORDS in include/yolo_hip.h


def conv2d_fwd_infer_unit(d, xp, wp, bias, epilogue, scale, shift, residual, out, absmax, pred2, in_bound, residual_bound,
                          planes, out_words, out_bound):
    """inference conv-BN-activation(-Add) unit whose result also leaves as planes (yolo_conv2d_fwd_infer_unit). in_bound /
    residual_bound: 1 float or the words another such unit left. Returns the number of words of out_words that now hold
    max|result| (one pass), or 0 when the bound is the float out_bound (two passes inside the call)."""
    _chk_f32(bias, scale, shift, residual, out, pred2, out_bound)
    if xp.numel() < planes_bytes(d.N * d.H * d.W, d.Cin) or wp.numel() < planes_bytes(d.Cout, d.kh * d.kw * d.Cin):
        raise YoloHipError("conv2d_fwd_infer_unit: planes buffers do not match the descriptor")
    if out.numel() != d.N * d.Ho * d.Wo * d.Cout or (residual is not None and residual.numel() != out.numel()):
        raise YoloHipError("conv2d_fwd_infer_unit: output / residual size does not match the descriptor")
    if planes.numel() < planes_bytes(d.N * d.Ho * d.Wo, d.Cout) or absmax.numel() < d.Cout or out_words.numel() < INFER_BOUND_WORDS:
        raise YoloHipError("conv2d_fwd_infer_unit: planes / absmax / out_words buffer too small")
    for b in (in_bound, residual_bound):
        if b is not None and (not 1 <= b.numel() <= INFER_BOUND_WORDS or b.element_size() != 4):
            raise YoloHipError("conv2d_fwd_infer_unit: a bound is 1..4096 four-byte words")
    n = ctypes.c_int(0)
    check(_lib.load().yolo_conv2d_fwd_infer_unit(byref(d), _p(xp), _p(wp), _p(bias), int(epilogue), _p(scale), _p(shift),
                                                 _p(residual), _p(out), _p(absmax), _p(pred2), _p(in_bound),
                                                 int(in_bound.numel()), _p(residual_bound),
                                                 int(residual_bound.numel()) if residual_bound is not None else 0,
                                                 _p(planes), _p(out_words), _p(out_bound), byref(n), _stream()),
          "yolo_conv2d_fwd_infer_unit")
    return int(n.value)


def absmax_words(x, words):
    """max|x| as one word per workgroup of the launch (yolo_absmax_words); returns their number"""
    _chk_f32(x)
    n = ctypes.c_int(0)
    check(_lib.load().yolo_absmax_words(_p(x), int(x.numel()), _p(words), byref(n), _stream()), "yolo_absmax_words")
    return int(n.value)


def stem_filter_prep(w, bias, wt):
    """wt (28 x 32 floats) <- the stem filter as the direct kernel reads it (yolo_stem_filter_prep)"""
    _chk_f32(w, bias, wt)
    if w.numel() != 32 * 27 or wt.numel() != 28 * 32:
        raise YoloHipError("stem_filter_prep: w must hold 32 x 27 floats, wt 28 x 32")
    check(_lib.load().yolo_stem_filter_prep(_p(w), _p(bias), _p(wt), _stream()), "yolo_stem_filter_prep")


def stem_infer_supported(d):
    return d.Cin == 3 and d.Cout == 32 and d.kh == 3 and d.kw == 3 and d.sh == 1 and d.sw == 1 and d.Ho == d.H and d.Wo == d.W


def stem_fwd_infer_unit(d, x, wt, epilogue, scale, shift, pred2, in_bound, out, planes, out_words):
    """the stem's inference unit in one launch (yolo_stem_fwd_infer_unit); out: fp32 result or None; returns the number of
    words of out_words that hold max|result|"""
    _chk_f32(x, wt, scale, shift, pred2, out)
    if planes.numel() < planes_bytes(d.N * d.H * d.W, 32) or out_words.numel() < INFER_BOUND_WORDS:
        raise YoloHipError("stem_fwd_infer_unit: planes / out_words buffer too small")
    if out is not None and out.numel() != d.N * d.H * d.W * 32:
        raise YoloHipError("stem_fwd_infer_unit: output size does not match the descriptor")
    n = ctypes.c_int(0)
    check(_lib.load().yolo_stem_fwd_infer_unit(byref(d), _p(x), _p(wt), int(epilogue), _p(scale), _p(shift), _p(pred2),
                                               _p(in_bound), int(in_bound.numel()), _p(out), _p(planes), _p(out_words),
                                               byref(n), _stream()), "yolo_stem_fwd_infer_unit")
    return int(n.value)


def fold_bound(words, out_bound):
    """out_bound[0] = max of 1..4096 non-negative floats (bit patterns): the words of a one-pass inference unit as one float"""
    _chk_f32(out_bound)
    check(_lib.load().yolo_fold_bound(_p(words), int(words.numel()), _p(out_bound), _stream()), "yolo_fold_bound")


def split_planes_absmax(x, rows, c, absmax, planes, extra_bound=None, out_bound=None):
    """planes of x [rows][c] with the bound max_c absmax[c] (+ extra_bound[0]); yolo_split_planes_absmax"""
    _chk_f32(x, extra_bound, out_bound)
    if planes.numel() < planes_bytes(rows, c):
        raise YoloHipError("split_planes_absmax: planes buffer too small")
    check(_lib.load().yolo_split_planes_absmax(_p(x), rows, c, _p(absmax), _p(extra_bound), _p(planes), _p(out_bound),
                                               _stream()), "yolo_split_planes_absmax")
    return planes


def split_planes_concat(srcs, channels, bounds, rows, planes, dst32=None, out_bound=None):
    """channel concatenation of up to four fp32 [rows][C_i] tensors straight into planes [rows][sum C_i]; bounds: one
    device float per source (>= max|source|); the result's bound -> out_bound (include/yolo_hip.h: yolo_split_planes_concat)"""
    n = len(srcs)
    _chk_f32(*srcs)
    C = sum(channels)
    if planes.numel() < planes_bytes(rows, C):
        raise YoloHipError("split_planes_concat: planes buffer too small")
    for t, c in zip(srcs, channels):
        if t.numel() != rows * c:
            raise YoloHipError("split_planes_concat: source size does not match rows x channels")
    xs = (c_void_p * n)(*[t.data_ptr() for t in srcs])
    bs = (c_void_p * n)(*[b.data_ptr() for b in bounds])
    cs = (ctypes.c_int * n)(*[int(c) for c in channels])
    if _tape.ACTIVE is not None:
        _tape.ACTIVE.keep.extend(list(srcs) + list(bounds))
    check(_lib.load().yolo_split_planes_concat(xs, cs, bs, n, int(rows), _p(planes), _p(dst32), _p(out_bound), _stream()),
          "yolo_split_planes_concat")
    return planes


def split_planes_concat_ex(srcs, channels, bounds, rows, planes, upsample=None, hw=(0, 0), dst32=None, out_bound=None):
    """split_planes_concat with (a) bounds given as WORDS -- bounds[i] is a device tensor of 1..4096 four-byte words whose
    maximum is the source's bound (one float, or what a one-pass inference unit left) -- and (b) sources read through
    UpSampling2D(2): upsample[i] true = srcs[i] is [N][H/2][W/2][channels[i]], hw = (H, W) of the result
    (include/yolo_hip.h: yolo_split_planes_concat_ex)"""
    n = len(srcs)
    _chk_f32(*srcs)
    C = sum(channels)
    up = [bool(f) for f in (upsample or [False] * n)]
    if planes.numel() < planes_bytes(rows, C):
        raise YoloHipError("split_planes_concat_ex: planes buffer too small")
    for t, c, f in zip(srcs, channels, up):
        if t.numel() != (rows // 4 if f else rows) * c:
            raise YoloHipError("split_planes_concat_ex: source size does not match rows x channels")
    for b in bounds:
        if b.dtype not in (torch.float32, torch.int32) or not 1 <= b.numel() <= INFER_BOUND_WORDS:
            raise YoloHipError("split_planes_concat_ex: a bound is 1..4096 four-byte words")
    xs = (c_void_p * n)(*[t.data_ptr() for t in srcs])
    bs = (c_void_p * n)(*[b.data_ptr() for b in bounds])
    cs = (ctypes.c_int * n)(*[int(c) for c in channels])
    bn = (ctypes.c_int * n)(*[int(b.numel()) for b in bounds])
    us = (ctypes.c_int * n)(*[int(f) for f in up])
    if _tape.ACTIVE is not None:
        _tape.ACTIVE.keep.extend(list(srcs) + list(bounds))
    check(_lib.load().yolo_split_planes_concat_ex(xs, cs, bs, bn, us, int(hw[0]), int(hw[1]), n, int(rows), _p(planes),
                                                  _p(dst32), _p(out_bound), _stream()), "yolo_split_planes_concat_ex")
    return planes


def conv2d_fwd_head_unit(d, xp, wp, bias, A, C, version, anchors_dev, t, y):
    """detection head: t = 1x1 convolution (+ bias) on planes operands, y = the head's activation of t; one launch where the
    launch has few output pixels (include/yolo_hip.h: yolo_conv2d_fwd_head_unit)"""
    _chk_f32(t, y)
    n = d.N * d.Ho * d.Wo * d.Cout
    if t.numel() != n or y.numel() != n:
        raise YoloHipError("conv2d_fwd_head_unit: output size does not match the descriptor")
    if xp.numel() < planes_bytes(d.N * d.H * d.W, d.Cin) or wp.numel() < planes_bytes(d.Cout, d.kh * d.kw * d.Cin):
        raise YoloHipError("conv2d_fwd_head_unit: planes buffers do not match the descriptor")
    check(_lib.load().yolo_conv2d_fwd_head_unit(ctypes.byref(d), _p(xp), _p(wp), _p(bias), int(A), int(C), int(version),
                                                _p(anchors_dev), _p(t), _p(y), _stream()), "yolo_conv2d_fwd_head_unit")
    return y


class BnReduce:
    """What yolo_conv2d_dgrad_planes_bnred needs to fold the BatchNormalization-backward reduction of the tensor it
    completes into its epilogue (include/yolo_hip.h): the pre-BN tensor y of the unit that PRODUCED the tensor, that unit's
    scale / shift / saved mean / saved 1/std and activation, a partials buffer ([cap][2][C] floats) and the 68 bound words.
    nslots is filled in by the launch."""
    __slots__ = ("y", "scale", "shift", "mean", "invstd", "act", "partials", "cap", "aux", "nslots")

    def __init__(self, y, scale, shift, mean, invstd, act, partials, cap, aux):
        self.y, self.scale, self.shift, self.mean, self.invstd, self.act = y, scale, shift, mean, invstd, act
        self.partials, self.cap, self.aux, self.nslots = partials, cap, aux, 0


def bnred_slots_cap(d):
    return int(_lib.load().yolo_bnred_slots_cap(byref(d)))


def conv2d_dgrad_planes(d, dyp, wTp, dx=None, accumulate=False, bnred=None):
    if dx is None:
        dx = torch.empty((d.N, d.H, d.W, d.Cin), device=dyp.device, dtype=torch.float32)
        accumulate = False
    if (dyp.numel() < planes_bytes(d.N * d.Ho * d.Wo, d.Cout) or wTp.numel() < planes_bytes(d.Cin, d.kh * d.kw * d.Cout)
            or dx.numel() != d.N * d.H * d.W * d.Cin):
        raise YoloHipError("conv2d_dgrad_planes: buffers do not match the descriptor")
    if bnred is not None:
        b = bnred
        _chk_f32(b.y, b.scale, b.shift, b.mean, b.invstd, b.partials)
        if (b.y.numel() != dx.numel() or min(t.numel() for t in (b.scale, b.shift, b.mean, b.invstd)) < d.Cin
                or b.partials.numel() < b.cap * 2 * d.Cin or b.aux is None or b.aux.numel() < 68):
            raise YoloHipError("conv2d_dgrad_planes: the fused reduction's buffers do not match the descriptor")
    def run():
        if bnred is None:
            check(_lib.load().yolo_conv2d_dgrad_planes(byref(d), _p(dyp), _p(wTp), _p(dx), int(bool(accumulate)), _stream()),
                  "yolo_conv2d_dgrad_planes")
            return
        n = ctypes.c_int(0)
        check(_lib.load().yolo_conv2d_dgrad_planes_bnred(byref(d), _p(dyp), _p(wTp), _p(dx), int(bool(accumulate)), _p(b.y),
                                                         _p(b.scale), _p(b.shift), _p(b.mean), _p(b.invstd), int(b.act),
                                                         _p(b.partials), int(b.cap), _p(b.aux), byref(n), _stream()),
              "yolo_conv2d_dgrad_planes_bnred")
        b.nslots = n.value
    if TIMER is not None:
        TIMER.bracket(_planes_variant(d.Cin, _win_key(d, d.Ho, d.Wo, d.Cout), d.kh * d.kw == 1 and d.sh * d.sw == 1), _conv_flops(d), 1, run,
                      _layer_key(d, "dgrad"))
    else:
        run()
    return dx


def conv2d_wgrad_planes(d, xp, dyp, dw, dy=None, dbias=None):
    """dw += filter gradient from pre-split x / dy planes; dbias (optional) needs the fp32 dy"""
    _chk_f32(dw, dy, dbias)
    if dw.numel() != d.Cout * d.kh * d.kw * d.Cin:
        raise YoloHipError("conv2d_wgrad_planes: dw size does not match the descriptor")
    if xp.numel() < planes_bytes(d.N * d.H * d.W, d.Cin) or dyp.numel() < planes_bytes(d.N * d.Ho * d.Wo, d.Cout):
        raise YoloHipError("conv2d_wgrad_planes: planes buffers do not match the descriptor")
    def run():
        check(_lib.load().yolo_conv2d_wgrad_planes(byref(d), _p(xp), _p(dyp), _p(dw), _stream()),
              "yolo_conv2d_wgrad_planes")
    if TIMER is not None:
        TIMER.bracket(_wgrad_planes_variant(d.Cout, d.kh * d.kw * d.Cin, d.kh * d.kw, d.N * d.Ho * d.Wo, d), _conv_flops(d), 1, run,
                      _layer_key(d, "wgrad"))
    else:
        run()
    if dbias is not None:
        if dy is None:
            raise YoloHipError("conv2d_wgrad_planes: dbias needs the fp32 dy")
        check(_lib.load().yolo_conv2d_wgrad_bias(_p(dy), d.N * d.Ho * d.Wo, d.Cout, _p(dbias), _stream()),
              "yolo_conv2d_wgrad_bias")
    return dw


def conv2d_wgrad_bias(dy, rows, c, dbias):
    """dbias += column sums of the fp32 dy [rows][c] (include/yolo_hip.h: yolo_conv2d_wgrad_bias)"""
    _chk_f32(dy, dbias)
    if dy.numel() != rows * c or dbias.numel() != c:
        raise YoloHipError("conv2d_wgrad_bias: sizes do not match")
    check(_lib.load().yolo_conv2d_wgrad_bias(_p(dy), int(rows), int(c), _p(dbias), _stream()), "yolo_conv2d_wgrad_bias")
    return dbias


def filter_transpose(w, cout, taps, cin, out=None):
    _chk_f32(w)
    if out is None:
        out = torch.empty(cin * taps * cout, device=w.device, dtype=torch.float32)
    check(_lib.load().yolo_filter_transpose(_p(w), _p(out), cout, taps, cin, _stream()), "yolo_filter_transpose")
    return out


def conv2d_dgrad(d, dy, wT, dx=None, accumulate=False):
    _chk_f32(dy, wT)
    if dx is None:
        dx = torch.empty((d.N, d.H, d.W, d.Cin), device=dy.device, dtype=torch.float32)
        accumulate = False
    if dy.numel() != d.N * d.Ho * d.Wo * d.Cout or dx.numel() != d.N * d.H * d.W * d.Cin:
        raise YoloHipError("conv2d_dgrad: tensor sizes do not match the descriptor")
    def run():
        check(_lib.load().yolo_conv2d_dgrad(byref(d), _p(dy), _p(wT), _p(dx), int(bool(accumulate)), _stream()),
              "yolo_conv2d_dgrad")
    if TIMER is not None:
        TIMER.bracket(_gather_variant(d.Cin, d.Cout % 32 != 0, d.N * (-(-d.H // d.sh)) * (-(-d.W // d.sw))), _conv_flops(d), d.sh * d.sw, run)
    else:
        run()
    return dx


def conv2d_wgrad(d, x, dy, dw, dbias=None):
    _chk_f32(x, dy, dw, dbias)
    if dw.numel() != d.Cout * d.kh * d.kw * d.Cin:
        raise YoloHipError("conv2d_wgrad: dw size does not match the descriptor")
    def run():
        check(_lib.load().yolo_conv2d_wgrad(byref(d), _p(x), _p(dy), _p(dw), None, _stream()), "yolo_conv2d_wgrad")
    if TIMER is not None:
        split = (CONV_MODE == "split" and d.Cout % 4 == 0 and d.Cin % 4 == 0 and d.Cout >= 64
                 and d.kh * d.kw * d.Cin >= 64)
        if split:  # mirrors launch_wgrad_split() in csrc/conv_wgrad_split.hip
            cols = d.kh * d.kw * d.Cin
            name = "wgrad_split_kernel<%d,%d,2,2>" % (64 if d.Cout <= 64 else 128, 64 if cols <= 64 else 128)
        else:
            name = "wgrad_kernel"
        TIMER.bracket(name, _conv_flops(d), 1, run)
    else:
        run()
    if dbias is not None:
        check(_lib.load().yolo_conv2d_wgrad_bias(_p(dy), d.N * d.Ho * d.Wo, d.Cout, _p(dbias), _stream()),
              "yolo_conv2d_wgrad_bias")
    return dw


def bn_stats(x, C, stats):
    P = x.numel() // C
    check(_lib.load().yolo_bn_stats(_p(x), P, C, _p(stats), _stream()), "yolo_bn_stats")


def bn_finalize(stats, P, C, gamma, beta, moving_mean, moving_var, scale, shift, save_mean, save_invstd,
                eps=BN_EPS, momentum=BN_MOMENTUM, unbiased=False, bound=None, absmax=None, mean_offset=None):
    """bound: optional int32 CUDA tensor (1 word, zeroed) that receives the bit pattern of an upper bound of
    max|act(BN(x))| (needed by bn_act_fwd(planes=...)); absmax: optional int32 [C] per-channel max|x| bit patterns
    from the conv epilogue (makes the bound tight); mean_offset: optional [C] conv bias that was left out of the
    convolution (it cancels in training-mode BN; only the moving mean adds it back)"""
    check(_lib.load().yolo_bn_finalize_offset(_p(stats), P, C, _p(gamma), _p(beta), eps, momentum, int(unbiased),
                                              _p(moving_mean), _p(moving_var), _p(scale), _p(shift), _p(save_mean),
                                              _p(save_invstd), _p(absmax), _p(bound), _p(mean_offset), _stream()),
          "yolo_bn_finalize")


def bn_infer_bound(C, scale, shift, absmax, bound):
    """inference: bound[0] (int32 view of a float) = max_c |scale_c| max|x_c| + |shift_c| >= max|act(scale x + shift)|"""
    check(_lib.load().yolo_bn_infer_bound(C, _p(scale), _p(shift), _p(absmax), _p(bound), _stream()), "yolo_bn_infer_bound")


def bn_fold_inference(C, gamma, beta, moving_mean, moving_var, scale, shift, eps=BN_EPS):
    check(_lib.load().yolo_bn_fold_inference(C, _p(gamma), _p(beta), _p(moving_mean), _p(moving_var), eps,
                                             _p(scale), _p(shift), _stream()), "yolo_bn_fold_inference")


def bn_act_fwd(x, C, scale, shift, act, residual=None, out=None, planes=None, bn_bound=None, residual_bound=None,
               out_bound=None, want_out=True, residual_planes=None):
    """planes: optional uint8 buffer (planes_bytes(P, C)) that also receives the result in the conv operand format;
    it needs bn_bound (from bn_finalize) and, with a residual, residual_bound (1 float: bound of the residual
    tensor). out_bound (1 float, optional) receives the bound of the result. want_out=False (with planes): only the
    planes are written."""
    if not want_out:
        if planes is None:
            raise YoloHipError("bn_act_fwd: nothing to produce")
        out = None
    elif out is None:
        out = torch.empty_like(x)
    P = x.numel() // C
    if planes is not None and planes.numel() < planes_bytes(P, C):
        raise YoloHipError("bn_act_fwd: planes buffer too small")
    if residual_planes is not None:   # the residual in the conv operand format (its fp32 copy need not exist)
        if residual is not None or residual_planes.numel() < planes_bytes(P, C):
            raise YoloHipError("bn_act_fwd: give the residual either as fp32 or as planes of the same shape")
        check(_lib.load().yolo_bn_act_fwd_res_planes(_p(x), P, C, _p(scale), _p(shift), act, _p(residual_planes), _p(out),
                                                     _p(planes), _p(bn_bound), _p(out_bound), _stream()),
              "yolo_bn_act_fwd_res_planes")
        return out
    check(_lib.load().yolo_bn_act_fwd_planes(_p(x), P, C, _p(scale), _p(shift), act, _p(residual), _p(out), _p(planes),
                                             _p(bn_bound), _p(residual_bound), _p(out_bound), _stream()),
          "yolo_bn_act_fwd")
    return out


class ChannelSlice:
    """dL/d(tensor) given as channels [c_off, c_off + C) of a wider dense tensor `base` [P][ld]: what the gradient of a
    Concatenate IS for each of its sources. bn_act_bwd reads it in place (row pitch ld); dense() copies it out for every other
    consumer."""
    __slots__ = ("base", "ld", "c_off", "C")

    def __init__(self, base, ld, c_off, C):
        _chk_f32(base)     # (contiguous: a reshape in view() would otherwise silently copy, and the row pitch would be wrong)
        if base.shape[-1] != ld or c_off < 0 or C <= 0 or c_off + C > ld:
            raise YoloHipError(f"ChannelSlice: channels [{c_off}, {c_off + C}) of a tensor whose last dimension is "
                               f"{base.shape[-1]} (row pitch given: {ld})")
        self.base, self.ld, self.c_off, self.C = base, ld, c_off, C

    def view(self):
        """flat view starting at the slice's first element (row p of the slice starts at p * ld)"""
        return self.base.reshape(-1)[self.c_off:]

    def dense(self, shape):
        out = torch.empty(shape, device=self.base.device, dtype=torch.float32)
        copy_channels_out(self.base, self.ld, self.c_off, out, self.C, accumulate=False)
        return out


BN_FOLD_TICKET_WORDS = 36   # include/yolo_hip.h YOLO_BN_FOLD_TICKET_WORDS


def bn_act_bwd(x, dout, C, gamma, scale, shift, save_mean, save_invstd, act, red, dgamma, dbeta, dx=None,
               planes=None, want_dx=True, bound_aux=None, fused=None, tickets=None):
    """returns dx (None when want_dx is False and only the planes of dx are produced). planes needs bound_aux:
    int32 CUDA tensor of 68 zeroed words (filled by the reduce step, read by the apply step). tickets: BN_FOLD_TICKET_WORDS
    int32 words, zero before their first use: the reduction then finishes inside its own launch (yolo_bn_act_bwd_reduce_fold_ld)
    instead of a second one. dout may be a ChannelSlice
    (C % 8 == 0): the two passes then read it in place with its row pitch. fused = the BnReduce whose data gradient already
    made the reduction of THIS dout (conv2d_dgrad_planes(bnred=...)): its slots are folded instead of a pass over x and dout."""
    ld = C
    if isinstance(dout, ChannelSlice):
        if dout.C != C or C % 8 != 0 or dout.c_off % 4 != 0 or dout.ld % 4 != 0:
            raise YoloHipError("bn_act_bwd: channel slice does not fit (C % 8, offset % 4, pitch % 4)")
        ld, dout = dout.ld, dout.view()
    P = x.numel() // C
    if planes is not None and planes.numel() < planes_bytes(P, C):
        raise YoloHipError("bn_act_bwd: planes buffer too small")
    if planes is not None and (bound_aux is None or bound_aux.numel() < 68):
        raise YoloHipError("bn_act_bwd: planes output needs bound_aux (68 zeroed int32 words)")
    if not want_dx and planes is None:
        raise YoloHipError("bn_act_bwd: nothing to produce")
    if dx is None and want_dx:
        dx = torch.empty_like(x)
    lib = _lib.load()
    if fused is not None:
        if fused.nslots <= 0 or fused.aux is None or bound_aux is None or fused.aux.data_ptr() != bound_aux.data_ptr():
            raise YoloHipError("bn_act_bwd: the fused reduction did not run for this tensor")
        check(lib.yolo_bn_act_bwd_sum_partials(_p(fused.partials), int(fused.nslots), P, C, _p(scale), _p(red), _p(bound_aux),
                                               _stream()), "yolo_bn_act_bwd_sum_partials")
    else:
        if tickets is not None and tickets.numel() < BN_FOLD_TICKET_WORDS:
            raise YoloHipError(f"bn_act_bwd: tickets needs {BN_FOLD_TICKET_WORDS} int32 words")
        check(lib.yolo_bn_act_bwd_reduce_fold_ld(_p(x), _p(dout), ld, P, C, _p(scale), _p(shift), _p(save_mean), _p(save_invstd),
                                                 act, _p(red), _p(bound_aux), _p(tickets), _stream()), "yolo_bn_act_bwd_reduce")
    check(lib.yolo_bn_act_bwd_apply_planes_ld(_p(x), _p(dout), ld, P, C, _p(gamma), _p(scale), _p(shift), _p(save_mean),
                                              _p(save_invstd), act, _p(red), _p(dgamma), _p(dbeta),
                                              _p(dx if want_dx else None), _p(planes), _p(bound_aux), _stream()),
          "yolo_bn_act_bwd_apply")
    return dx if want_dx else None


_STEM_SCRATCH = None


def stem_bn_bwd_supported(d):
    """the fused stem backward (csrc/stem.hip): Conv2D(32, 3x3, stride 1, 'same') on 3 channels"""
    return (d.Cin == 3 and d.Cout == 32 and d.kh == 3 and d.kw == 3 and d.sh == 1 and d.sw == 1 and d.Ho == d.H
            and d.Wo == d.W and d.W >= 16)


def stem_bn_bwd_wgrad(d, image, y, dout, scale, shift, save_mean, save_invstd, act, red, dgamma, dbeta, dw, fused=None):
    """backward of the stem unit in one pass over (y, dout): yolo_bn_act_bwd_reduce (or, fused = the BnReduce of the data
    gradient that completed dout, the fold of its slots), then the apply step fused with the filter gradient
    (yolo_stem_bn_bwd_wgrad); dw / dgamma / dbeta are accumulated"""
    global _STEM_SCRATCH
    _chk_f32(image, y, dout, scale, shift, save_mean, save_invstd, dgamma, dbeta, dw)
    lib = _lib.load()
    P = d.N * d.H * d.W
    if image.numel() != P * 3 or y.numel() != P * 32 or dout.numel() != P * 32 or dw.numel() != 32 * 27:
        raise YoloHipError("stem_bn_bwd_wgrad: tensor sizes do not match the descriptor")
    if _STEM_SCRATCH is None or _STEM_SCRATCH.device != y.device:
        _STEM_SCRATCH = torch.empty(int(lib.yolo_stem_bwd_scratch_bytes()), dtype=torch.uint8, device=y.device)
    if fused is not None:
        if fused.nslots <= 0:
            raise YoloHipError("stem_bn_bwd_wgrad: the fused reduction did not run for this tensor")
        check(lib.yolo_bn_act_bwd_sum_partials(_p(fused.partials), int(fused.nslots), P, 32, _p(scale), _p(red), _p(fused.aux),
                                               _stream()), "yolo_bn_act_bwd_sum_partials")
    else:
        check(lib.yolo_bn_act_bwd_reduce_bound(_p(y), _p(dout), P, 32, _p(scale), _p(shift), _p(save_mean), _p(save_invstd),
                                               act, _p(red), _p(None), _stream()), "yolo_bn_act_bwd_reduce")
    def run():
        check(lib.yolo_stem_bn_bwd_wgrad(byref(d), _p(y), _p(dout), _p(image), _p(scale), _p(shift), _p(save_mean),
                                         _p(save_invstd), act, _p(red), _p(dgamma), _p(dbeta), _p(dw), _p(_STEM_SCRATCH),
                                         _STEM_SCRATCH.numel(), _stream()), "yolo_stem_bn_bwd_wgrad")
    if TIMER is not None:
        TIMER.bracket("stem_bn_bwd_wgrad_kernel", _conv_flops(d), 1, run)
    else:
        run()


def act_fwd(x, act, out=None):
    if out is None:
        out = torch.empty_like(x)
    check(_lib.load().yolo_act_fwd(_p(x), x.numel(), act, _p(out), _stream()), "yolo_act_fwd")
    return out


def act_bwd(x, dout, act, dx=None):
    if dx is None:
        dx = torch.empty_like(x)
    check(_lib.load().yolo_act_bwd(_p(x), _p(dout), x.numel(), act, _p(dx), _stream()), "yolo_act_bwd")
    return dx


def copy_channels_in(src, Csrc, dst, Cdst, c_off):
    P = src.numel() // Csrc
    if dst.numel() != P * Cdst:
        raise YoloHipError("copy_channels_in: pixel counts differ")
    check(_lib.load().yolo_copy_channels_in(_p(src), P, Csrc, _p(dst), Cdst, c_off, _stream()),
          "yolo_copy_channels_in")


def copy_channels_out(src, Csrc, c_off, dst, Cdst, accumulate=False):
    P = src.numel() // Csrc
    if dst.numel() != P * Cdst:
        raise YoloHipError("copy_channels_out: pixel counts differ")
    check(_lib.load().yolo_copy_channels_out(_p(src), P, Csrc, c_off, _p(dst), Cdst, int(bool(accumulate)),
                                             _stream()), "yolo_copy_channels_out")


def upsample2x_fwd(x, y, Cy, c_off):
    n, h, w, c = x.shape
    if y.numel() != n * 4 * h * w * Cy:
        raise YoloHipError("upsample2x_fwd: bad output size")
    check(_lib.load().yolo_upsample2x_fwd(_p(x), n, h, w, c, _p(y), Cy, c_off, _stream()), "yolo_upsample2x_fwd")


def upsample2x_bwd(dy, Cy, c_off, dx, accumulate=False):
    n, h, w, c = dx.shape
    if dy.numel() != n * 4 * h * w * Cy:
        raise YoloHipError("upsample2x_bwd: bad input size")
    check(_lib.load().yolo_upsample2x_bwd(_p(dy), n, h, w, c, Cy, c_off, _p(dx), int(bool(accumulate)), _stream()),
          "yolo_upsample2x_bwd")


def axpy(a, b):
    if a.numel() != b.numel():
        raise YoloHipError("axpy: size mismatch")
    check(_lib.load().yolo_axpy(_p(a), _p(b), a.numel(), _stream()), "yolo_axpy")


def fill(t, value):
    check(_lib.load().yolo_fill(_p(t), t.numel(), float(value), _stream()), "yolo_fill")


def zero_bytes(t):
    """all-zero bit pattern over a contiguous tensor of any 4- or 8-byte dtype (the library's own fill kernel: no torch
    operator runs on the product path)"""
    n = t.numel() * t.element_size()
    if n % 4:
        raise YoloHipError("zero_bytes: size must be a multiple of 4 bytes")
    if n:
        check(_lib.load().yolo_fill(_p(t), n // 4, 0.0, _stream()), "yolo_fill")


def maxpool_fwd(x, k, s, pad_t, pad_l, Ho, Wo, y, Cy, c_off, argmax):
    n, h, w, c = x.shape
    if y.numel() != n * Ho * Wo * Cy or (argmax is not None and argmax.numel() != n * Ho * Wo * c):
        raise YoloHipError("maxpool_fwd: bad output size")
    check(_lib.load().yolo_maxpool_fwd(_p(x), n, h, w, c, k, s, pad_t, pad_l, Ho, Wo, _p(y), Cy, c_off,
                                       _p(argmax), _stream()), "yolo_maxpool_fwd")


def maxpool_bwd(dy, N, Ho, Wo, C, Cy, c_off, argmax, dx):
    check(_lib.load().yolo_maxpool_bwd(_p(dy), N, Ho, Wo, C, Cy, c_off, _p(argmax), _p(dx), _stream()),
          "yolo_maxpool_bwd")


def bn_act_maxpool2x2_fwd(y, C, scale, shift, act, argmax, out=None, planes=None, bn_bound=None, out_bound=None):
    """BatchNormalization + activation + MaxPooling2D(2, 2) of the conv output y [N, 2 Ho, 2 Wo, C] in one pass
    (yolo_bn_act_maxpool2x2_fwd): fp32 pooled tensor `out` and / or its planes, argmax as yolo_maxpool_fwd records it"""
    n, h, w, c = y.shape
    if c != C or h % 2 or w % 2 or C % 8 or argmax.numel() != n * (h // 2) * (w // 2) * C:
        raise YoloHipError("bn_act_maxpool2x2_fwd: y must be [N, even, even, C % 8 == 0], argmax [N, H/2, W/2, C]")
    if out is None and planes is None:
        raise YoloHipError("bn_act_maxpool2x2_fwd: nothing to produce")
    if planes is not None and planes.numel() < planes_bytes(n * (h // 2) * (w // 2), C):
        raise YoloHipError("bn_act_maxpool2x2_fwd: planes buffer too small")
    check(_lib.load().yolo_bn_act_maxpool2x2_fwd(_p(y), n, h // 2, w // 2, C, _p(scale), _p(shift), int(act), _p(out), _p(argmax),
                                                 _p(planes), _p(bn_bound), _p(out_bound), _stream()),
          "yolo_bn_act_maxpool2x2_fwd")
    return out


def maxpool2x2_bwd(dy, N, Ho, Wo, C, argmax, dx, accumulate):
    """backward of a 2x2 / stride-2 pool whose windows tile its [N, 2 Ho, 2 Wo, C] input: writes (accumulate False: no zero
    fill needed) or adds to every input position exactly once, without atomics"""
    if dx.numel() != N * 4 * Ho * Wo * C or dy.numel() != N * Ho * Wo * C or argmax.numel() != dy.numel():
        raise YoloHipError("maxpool2x2_bwd: sizes do not match a pool that tiles its input")
    check(_lib.load().yolo_maxpool2x2_bwd(_p(dy), N, Ho, Wo, C, _p(argmax), _p(dx), 1 if accumulate else 0, _stream()),
          "yolo_maxpool2x2_bwd")


def maxpool_bwd_same(dy, N, H, W, C, Cy, c_off, argmax, k, pad_t, pad_l, dx):
    """backward of a stride-1 'same' pool without atomics (yolo_maxpool_bwd_same): dx += gathered dy"""
    check(_lib.load().yolo_maxpool_bwd_same(_p(dy), N, H, W, C, Cy, c_off, _p(argmax), int(k), int(pad_t), int(pad_l), _p(dx),
                                            _stream()), "yolo_maxpool_bwd_same")


def space_to_depth2_fwd(x, y, Cy, c_off):
    n, h, w, c = x.shape
    check(_lib.load().yolo_space_to_depth2_fwd(_p(x), n, h, w, c, _p(y), Cy, c_off, _stream()),
          "yolo_space_to_depth2_fwd")


def space_to_depth2_bwd(dy, Cy, c_off, dx, accumulate=False):
    n, h, w, c = dx.shape
    check(_lib.load().yolo_space_to_depth2_bwd(_p(dy), n, h, w, c, Cy, c_off, _p(dx), int(bool(accumulate)),
                                               _stream()), "yolo_space_to_depth2_bwd")


def head_act_fwd(t, A, C, version, anchors_dev, out=None):
    if out is None:
        out = torch.empty_like(t)
    D = (5 * A + C) if version == 1 else A * (5 + C)
    P = t.numel() // D
    check(_lib.load().yolo_head_act_fwd(_p(t), P, A, C, version, _p(anchors_dev), _p(out), _stream()),
          "yolo_head_act_fwd")
    return out


def head_act_bwd(y, dy, A, C, version, anchors_dev, dt=None, danchors=None):
    if dt is None:
        dt = torch.empty_like(y)
    D = (5 * A + C) if version == 1 else A * (5 + C)
    P = y.numel() // D
    check(_lib.load().yolo_head_act_bwd(_p(y), _p(dy), P, A, C, version, _p(anchors_dev), _p(dt), _p(danchors),
                                        _stream()), "yolo_head_act_bwd")
    return dt


def make_loss_cfg(version, N, gh, gw, A, C, anchors=None, binary_weight=1.0, loss_weight=(1, 1, 1, 1),
                  ignore_thresh=0.6, use_focal_loss=False, focal_gamma=2.0, use_scale=True, wh_reg_weight=0.01,
                  truth_thresh=1.0, label_smooth=0.0):
    cfg = LossCfg()
    cfg.version, cfg.N, cfg.gh, cfg.gw, cfg.A, cfg.C = version, N, gh, gw, A, C
    if anchors is not None:
        flat = [float(v) for pair in anchors for v in pair]
        if len(flat) != 2 * A or len(flat) > 32:
            raise ValueError(f"need {A} (w,h) anchors, at most 16; got {len(flat) // 2}")
        for i, v in enumerate(flat):
            cfg.anchors[i] = v
        cfg.use_anchors = 1
    else:
        cfg.use_anchors = 0
    cfg.binary_weight = float(binary_weight)
    lw = list(loss_weight) + [0.0] * (4 - len(loss_weight))
    for i in range(4):
        cfg.loss_weight[i] = float(lw[i])
    cfg.ignore_thresh = float(ignore_thresh)
    cfg.use_focal_loss = int(bool(use_focal_loss))
    cfg.focal_gamma = float(focal_gamma)
    cfg.use_scale = int(bool(use_scale))
    cfg.wh_reg_weight = float(wh_reg_weight)
    cfg.truth_thresh = float(truth_thresh)
    cfg.label_smooth = float(label_smooth)
    return cfg


def loss_fwd_bwd(cfg, y_true, y_pred, loss_out=None, dpred=None, grad_scale=1.0, want_grad=True, decisions=None):
    """decisions (optional): int32 CUDA tensor [cells, 2] that receives every cell's responsible anchor and its ignore /
    truth mask bits as this execution decided them (include/yolo_hip.h; parity tests)"""
    _chk_f32(y_true, y_pred, dpred)
    cells = cfg.N * cfg.gh * cfg.gw
    pd = (5 * cfg.A + cfg.C) if cfg.version == 1 else cfg.A * (5 + cfg.C)
    if y_true.numel() != cells * (5 + cfg.C) or y_pred.numel() != cells * pd:
        raise YoloHipError(f"loss: tensor sizes do not match cfg (cells={cells}, C={cfg.C}, A={cfg.A})")
    if loss_out is None:
        loss_out = torch.empty(8, device=y_pred.device, dtype=torch.float64)
    if want_grad and dpred is None:
        dpred = torch.empty_like(y_pred)
    if decisions is not None and (decisions.dtype != torch.int32 or decisions.numel() < 2 * cells or not decisions.is_cuda):
        raise YoloHipError("loss: decisions must be an int32 CUDA tensor with 2 entries per cell")
    check(_lib.load().yolo_loss_fwd_bwd(byref(cfg), _p(y_true), _p(y_pred), _p(loss_out),
                                        _p(dpred if want_grad else None), float(grad_scale), _p(decisions),
                                        0 if decisions is None else decisions.numel() * 4, _stream()), "yolo_loss_fwd_bwd")
    return loss_out, dpred


def metrics(cfg, y_true, y_pred, recall_thresh=0.5, out=None):
    _chk_f32(y_true, y_pred)
    if out is None:
        out = torch.empty(8, device=y_pred.device, dtype=torch.float64)
    check(_lib.load().yolo_metrics(byref(cfg), _p(y_true), _p(y_pred), float(recall_thresh), _p(out), _stream()),
          "yolo_metrics")
    return out


def encode_labels(boxes, cls, first, n_images, img_hw, grid_shape, class_num):
    """yolo_encode_labels: boxes float64 [nb,4] (x1,y1,x2,y2 pixels), cls int32 [nb], first int32 [N+1] (CUDA tensors).
    Returns (label64 [N,gh,gw,5+C] float64, label32 float32)."""
    if boxes.dtype != torch.float64 or cls.dtype != torch.int32 or first.dtype != torch.int32:
        raise YoloHipError("encode_labels: boxes float64, cls / first int32")
    gh, gw = grid_shape
    l64 = torch.empty((n_images, gh, gw, 5 + class_num), device=boxes.device, dtype=torch.float64)
    l32 = torch.empty((n_images, gh, gw, 5 + class_num), device=boxes.device, dtype=torch.float32)
    check(_lib.load().yolo_encode_labels(_p(boxes.contiguous()), _p(cls.contiguous()), _p(first.contiguous()), n_images,
                                         float(img_hw[0]), float(img_hw[1]), gh, gw, class_num, _p(l64), _p(l32),
                                         _stream()), "yolo_encode_labels")
    return l64, l32


def down2xlabel(label64):
    """yolo_down2xlabel on a float64 CUDA label [N,gh,gw,ch] -> (float64, float32) labels of the 2x coarser grid"""
    if label64.dtype != torch.float64 or not label64.is_cuda or not label64.is_contiguous():
        raise YoloHipError("down2xlabel: contiguous float64 CUDA tensor expected")
    n, gh, gw, ch = label64.shape
    o64 = torch.empty((n, gh // 2, gw // 2, ch), device=label64.device, dtype=torch.float64)
    o32 = torch.empty((n, gh // 2, gw // 2, ch), device=label64.device, dtype=torch.float32)
    check(_lib.load().yolo_down2xlabel(_p(label64), n, gh, gw, ch, _p(o64), _p(o32), _stream()), "yolo_down2xlabel")
    return o64, o32


def adam_step(p, g, m, v, lr, step, beta1=0.9, beta2=0.999, eps=1e-7, grad_scale=1.0, zero_grad=True):
    check(_lib.load().yolo_adam_step(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, int(step),
                                     float(grad_scale), int(bool(zero_grad)), _stream()), "yolo_adam_step")


def adam_lr_t(lr, step, beta1=0.9, beta2=0.999):
    return float(_lib.load().yolo_adam_lr_t(float(lr), float(beta1), float(beta2), int(step)))


def adam_step_dev(p, g, m, v, hyper, zero_grad=True):
    """Adam with {lr_t, beta1, beta2, eps, grad_scale} read from the device tensor `hyper` (float32[5]): the capturable form"""
    check(_lib.load().yolo_adam_step_dev(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper), int(zero_grad), _stream()),
          "yolo_adam_step_dev")


def sgd_step(p, g, lr, grad_scale=1.0, zero_grad=True):
    check(_lib.load().yolo_sgd_step(_p(p), _p(g), p.numel(), lr, float(grad_scale), int(bool(zero_grad)), _stream()),
          "yolo_sgd_step")


def decode_level(pred, A, C, version, threshold, rows, max_rows, count, workspace):
    gh, gw = pred.shape[0], pred.shape[1]
    lib = _lib.load()
    if pred.dtype == torch.float32:
        check(lib.yolo_decode_level(_p(pred), gh, gw, A, C, version, float(threshold), _p(rows), max_rows,
                                    _p(count), _p(workspace), workspace.numel() * workspace.element_size(),
                                    _stream()), "yolo_decode_level")
    elif pred.dtype == torch.float64:
        check(lib.yolo_decode_level_f64(_p(pred), gh, gw, A, C, version, float(threshold), _p(rows), max_rows,
                                        _p(count), _p(workspace), workspace.numel() * workspace.element_size(),
                                        _stream()), "yolo_decode_level_f64")
    else:
        raise YoloHipError(f"decode: unsupported dtype {pred.dtype}")


def decode_workspace_bytes(gh, gw, A, C):
    return int(_lib.load().yolo_decode_workspace_bytes(gh, gw, A, C))


def nms_keep(rows, class_num, mode, nms_threshold, conf_threshold=0.5, sigma=0.5):
    """rows: float64 CUDA tensor [n,7]; returns uint8 keep mask [n]."""
    n = rows.shape[0]
    keep = torch.empty(n, device=rows.device, dtype=torch.uint8)
    if n == 0:
        return keep
    lib = _lib.load()
    nbytes = int(lib.yolo_nms_workspace_bytes(n, class_num))
    ws = torch.empty(nbytes, device=rows.device, dtype=torch.uint8)
    check(lib.yolo_nms(_p(rows), n, class_num, mode, float(nms_threshold), float(conf_threshold), float(sigma),
                       _p(keep), _p(ws), nbytes, _stream()), "yolo_nms")
    return keep


def nms_select(rows, class_num, mode, nms_threshold, conf_threshold=0.5, sigma=0.5):
    """rows: float64 CUDA tensor [n,7]; returns the kept rows in the reference's output order (classes ascending, original
    order inside a class) as a float64 CUDA tensor [m,7] -- NMS and the ordered gather in one call, one host sync (m)."""
    n = rows.shape[0]
    if n == 0:
        return rows.reshape(0, 7)
    lib = _lib.load()
    nbytes = int(lib.yolo_nms_workspace_bytes(n, class_num))
    ws = torch.empty(nbytes, device=rows.device, dtype=torch.uint8)
    keep = torch.empty(n, device=rows.device, dtype=torch.uint8)
    out = torch.empty((n, 7), device=rows.device, dtype=torch.float64)
    count = torch.empty(1, device=rows.device, dtype=torch.int32)
    check(lib.yolo_nms_select(_p(rows), n, class_num, mode, float(nms_threshold), float(conf_threshold), float(sigma),
                              _p(keep), _p(out), _p(count), _p(ws), nbytes, _stream()), "yolo_nms_select")
    return out[:int(count.item())]


# ---- evaluation after decode / NMS (csrc/measure.hip) ---------------------------------------------------
def match_detections(gt_rows, det_rows, class_num, iou_threshold, class_counts):
    """gt_rows [ngt,7], det_rows [ndet,7] float64 CUDA tensors of ONE image; class_counts int64 CUDA tensor
    [class_num,4] (accumulated). Returns (best_gt int32 [ndet], matched uint8 [ndet], best_iou float64 [ndet])."""
    dev = class_counts.device
    ngt, ndet = int(gt_rows.shape[0]), int(det_rows.shape[0])
    if ((ngt and gt_rows.dtype != torch.float64) or (ndet and det_rows.dtype != torch.float64)
            or class_counts.dtype != torch.int64 or class_counts.numel() != class_num * 4):
        raise YoloHipError("match_detections: rows must be float64 [n,7], class_counts int64 [class_num,4]")
    best_gt = torch.empty(ndet, device=dev, dtype=torch.int32)
    matched = torch.empty(ndet, device=dev, dtype=torch.uint8)
    best_iou = torch.empty(ndet, device=dev, dtype=torch.float64)
    flags = torch.empty(max(ngt, 1), device=dev, dtype=torch.uint8)
    check(_lib.load().yolo_match_detections(_p(gt_rows.contiguous()) if ngt else None, ngt,
                                            _p(det_rows.contiguous()) if ndet else None, ndet, class_num,
                                            float(iou_threshold), _p(best_gt), _p(matched), _p(best_iou), _p(flags),
                                            _p(class_counts), _stream()), "yolo_match_detections")
    return best_gt, matched, best_iou


def rank_desc(key, segment=None):
    """rank of every element inside its segment, keys descending (ties: later element first)"""
    n = int(key.numel())
    rank = torch.empty(n, device=key.device, dtype=torch.int32)
    if n:
        if key.dtype != torch.float64 or (segment is not None and segment.dtype != torch.int32):
            raise YoloHipError("rank_desc: key must be float64, segment int32")
        check(_lib.load().yolo_rank_desc(_p(key.contiguous()), _p(segment), n, _p(rank), _stream()), "yolo_rank_desc")
    return rank


def pr_curve(joint, gt_id, matched, num_gts, precision_mode):
    """one class: (precision, recall) float64 CUDA tensors of n+1 points"""
    n = int(joint.numel())
    if joint.dtype != torch.float64 or gt_id.dtype != torch.int32 or matched.dtype != torch.uint8:
        raise YoloHipError("pr_curve: joint float64, gt_id int32, matched uint8 expected")
    lib = _lib.load()
    nbytes = int(lib.yolo_pr_curve_workspace_bytes(n, int(num_gts)))
    ws = torch.empty(nbytes, device=joint.device, dtype=torch.uint8)
    prec = torch.empty(n + 1, device=joint.device, dtype=torch.float64)
    rec = torch.empty(n + 1, device=joint.device, dtype=torch.float64)
    check(lib.yolo_pr_curve(_p(joint.contiguous()), _p(gt_id.contiguous()), _p(matched.contiguous()), n, int(num_gts),
                            int(precision_mode), _p(ws), nbytes, _p(prec), _p(rec), _stream()), "yolo_pr_curve")
    return prec, rec


# ---- stand-alone cal_iou (csrc/iou.hip) -------------------------------------------------------------------
def cal_iou(xywh_true, xywh_pred, mode=1, grid_wh=(1.0, 1.0)):
    """Broadcasting IoU (mode 1) / DIoU (2) / (IoU, CIoU) (3) of boxes [..., >=4] (x, y, w, h first); both CUDA tensors of
    ONE float dtype (float32 or float64). The leading dimensions broadcast like NumPy's; nothing is materialised:
    the kernel walks the operands through their strides. Returns a tensor of the broadcast shape (two for mode 3)."""
    import ctypes
    if xywh_true.dtype != xywh_pred.dtype or xywh_true.dtype not in (torch.float32, torch.float64):
        raise YoloHipError("cal_iou: operands must share one dtype, float32 or float64")
    if xywh_true.shape[-1] < 4 or xywh_pred.shape[-1] < 4:
        raise YoloHipError("cal_iou: the last dimension holds x, y, w, h")
    for t in (xywh_true, xywh_pred):
        if t.numel() and t.stride(-1) != 1:
            raise YoloHipError("cal_iou: x, y, w, h must be consecutive in memory")
    lead_a, lead_b = tuple(xywh_true.shape[:-1]), tuple(xywh_pred.shape[:-1])
    shape = tuple(torch.broadcast_shapes(lead_a, lead_b))
    nd = len(shape)
    if nd > 8:
        raise YoloHipError("cal_iou: at most 8 leading dimensions")

    def strides(t, lead):
        out = [0] * nd
        for i in range(len(lead)):
            d = nd - len(lead) + i
            out[d] = 0 if lead[i] == 1 and shape[d] != 1 else int(t.stride(i))
        return out
    arr = ctypes.c_longlong * max(nd, 1)
    out = torch.empty(shape, device=xywh_true.device, dtype=xywh_true.dtype)
    out2 = torch.empty(shape, device=xywh_true.device, dtype=xywh_true.dtype) if mode == 3 else None
    if out.numel() == 0:
        return (out, out2) if mode == 3 else out
    check(_lib.load().yolo_cal_iou(_p(xywh_true), _p(xywh_pred), _p(out), _p(out2),
                                   1 if xywh_true.dtype == torch.float64 else 0, int(mode), nd, arr(*shape),
                                   arr(*strides(xywh_true, lead_a)), arr(*strides(xywh_pred, lead_b)),
                                   float(grid_wh[0]), float(grid_wh[1]), _stream()), "yolo_cal_iou")
    return (out, out2) if mode == 3 else out


# ---- roofline yardstick (csrc/probe.hip) --------------------------------------------------------------------
def mfma_ceiling(seconds=0.15, workgroups=512):
    """Raw fp16 MFMA TFLOP/s this chip sustains on random register operands with nothing else in the loop
    (yolo_mfma_probe, timed with events on the current stream): the held-clock ceiling of bench.py's roofline block.
    Runs a short calibration launch, then ONE launch sized for about `seconds`."""
    import ctypes
    lib = _lib.load()
    n = workgroups * 512
    g = torch.Generator(device="cuda").manual_seed(1)
    ops_ = (torch.rand(n * 32, device="cuda", generator=g) * 2 - 1).to(torch.float16)
    sink = torch.empty(n, device="cuda", dtype=torch.float32)
    fl = ctypes.c_double(0.0)

    def run(iters):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        check(lib.yolo_mfma_probe(_p(ops_), _p(sink), workgroups, iters, ctypes.byref(fl), _stream()), "yolo_mfma_probe")
        e.record()
        e.synchronize()
        return s.elapsed_time(e) * 1e-3, fl.value
    run(200)
    t, f = run(2000)
    iters = max(2000, int(2000 * seconds / max(t, 1e-6)))
    t, f = run(iters)
    return {"raw_fp16_mfma_tflops": round(f / t / 1e12, 1), "seconds": round(t, 4), "workgroups": workgroups, "iters": iters}
