"""GPU parity of the implicit-GEMM conv kernels (fwd / dgrad / wgrad) against the torch-CPU
float64 oracle (oracle/layers.py, Keras padding semantics). Tolerance: fp32 1e-4 relative to
the tensor scale (BASELINE.json north_star)."""
import os
import pytest
import torch

from oracle import layers as L

pytestmark = pytest.mark.gpu

TOL = 1e-4

# (N, H, W, Cin, Cout, k, stride, padding, bias)
CASES = [
    (2, 16, 16, 32, 64, 3, 1, "same", False),       # plain 3x3
    (2, 16, 16, 64, 32, 1, 1, "same", False),       # 1x1, Cout = 32 tile
    (2, 17, 13, 32, 64, 3, 2, "darknet_s2", False),  # v3 down-sampling, odd sizes
    (2, 16, 16, 32, 128, 3, 2, "darknet_s2", False),
    (1, 20, 20, 3, 32, 3, 1, "same", False),        # Cin = 3 stem (flat-K path)
    (2, 14, 14, 64, 64, 3, 2, "same", True),        # v1: 3x3 s2 'same' (asymmetric pad), bias
    (1, 28, 28, 3, 64, 7, 2, "same", True),         # v1 stem 7x7 s2 'same'
    (2, 13, 13, 128, 255, 1, 1, "same", True),      # v3 head: Cout = 255
    (2, 7, 7, 96, 160, 3, 1, "same", False),        # Cin multiple of 32 but not of 64; M tail
    (3, 9, 11, 256, 128, 1, 1, "valid", False),
    (1, 13, 13, 64, 125, 1, 1, "same", True),       # v2 head width
    (1, 5, 5, 1024, 11, 1, 1, "same", True),        # v1 head (tiny Cout)
    (3, 21, 19, 3, 32, 3, 1, "same", True),         # stem, small: implicit-GEMM path
    (2, 733, 717, 3, 32, 3, 1, "same", True),       # stem >= 2^20 pixels: direct kernel (csrc/stem.hip), odd sizes
]


def _mk(case, seed=0):
    n, h, w, cin, cout, k, s, pad, bias = case
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(n, h, w, cin, generator=g, dtype=torch.float64)
    wk = torch.randn(k, k, cin, cout, generator=g, dtype=torch.float64) / (k * k * cin) ** 0.5
    b = torch.randn(cout, generator=g, dtype=torch.float64) if bias else None
    return x, wk, b


def _krsc(w_hwio):
    return w_hwio.permute(3, 0, 1, 2).contiguous()


def _relerr(a, b):
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


@pytest.mark.parametrize("case", CASES)
def test_conv_fwd(case):
    from tf2_yolo_amd import ops
    n, h, w, cin, cout, k, s, pad, bias = case
    x, wk, b = _mk(case)
    ref = L.conv2d(x, wk, b, stride=s, padding=pad)
    d = ops.conv_desc((n, h, w, cin), cout, k, k, s, pad)
    assert (d.Ho, d.Wo) == (ref.shape[1], ref.shape[2])
    y = ops.conv2d_fwd(d, x.float().cuda(), _krsc(wk).float().cuda(), None if b is None else b.float().cuda())
    torch.cuda.synchronize()
    assert _relerr(y.double().cpu(), ref) < TOL


@pytest.mark.parametrize("case", CASES)
def test_conv_dgrad_wgrad(case):
    from tf2_yolo_amd import ops
    n, h, w, cin, cout, k, s, pad, bias = case
    x, wk, b = _mk(case, seed=1)
    x.requires_grad_(True)
    wk.requires_grad_(True)
    if b is not None:
        b.requires_grad_(True)
    ref = L.conv2d(x, wk, b, stride=s, padding=pad)
    g = torch.Generator().manual_seed(2)
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(dy)
    d = ops.conv_desc((n, h, w, cin), cout, k, k, s, pad)
    wd = _krsc(wk.detach()).float().cuda()
    dyd = dy.float().cuda()
    # wgrad (+ bias grad)
    dw = torch.zeros_like(wd)
    db = torch.zeros(cout, device="cuda") if bias else None
    ops.conv2d_wgrad(d, x.detach().float().cuda(), dyd, dw, db)
    torch.cuda.synchronize()
    assert _relerr(dw.double().cpu(), _krsc(wk.grad)) < TOL
    if bias:
        assert _relerr(db.double().cpu(), b.grad) < TOL
    # dgrad (not defined for the flat-K stem in the product: the image has no gradient)
    if cin % 32 == 0 and (cout % 32 == 0 or k == 1):
        wT = ops.filter_transpose(wd, cout, k * k, cin)
        dx = ops.conv2d_dgrad(d, dyd, wT)
        torch.cuda.synchronize()
        assert _relerr(dx.double().cpu(), x.grad) < TOL
        # accumulate form: dx += ...
        dx2 = dx.clone()
        ops.conv2d_dgrad(d, dyd, wT, dx=dx2, accumulate=True)
        torch.cuda.synchronize()
        assert _relerr(dx2.double().cpu(), 2 * x.grad) < TOL


@pytest.mark.parametrize("case", [CASES[6], CASES[4], CASES[11], CASES[7], (4, 112, 112, 3, 64, 7, 2, "same", False),
                                  (2, 40, 40, 24, 40, 3, 1, "same", True)])
def test_conv_wgrad_exact_kernel_reproducible(case):
    """yolo_conv2d_wgrad on fp32 operands (the 7x7 RGB stem of YOLOv1.5, heads narrower than 32 filters, channel counts the
    planes kernels do not take): with the wgrad workspace registered the splits of the pixel contraction go to slabs and are
    added IN ORDER (wgrad_exact_reduce_kernel) instead of meeting in fp32 atomics -- until round 6 this kernel's gradients
    differed in the last bit from run to run (scripts/step_repro.py c1: conv1_conv/kernel after ONE step). Two runs
    bit-identical, accumulating onto dw, equal to the atomic form to summation order and to the float64 oracle to 1e-4."""
    from tf2_yolo_amd import _lib, ops
    n, h, w, cin, cout, k, s, pad, bias = case
    x, wk, b = _mk(case, seed=41)
    wk.requires_grad_(True)
    ref = L.conv2d(x, wk, None, stride=s, padding=pad)
    g = torch.Generator().manual_seed(42)
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(dy)
    d = ops.conv_desc((n, h, w, cin), cout, k, k, s, pad)
    xd, dyd = x.float().cuda(), dy.float().cuda()

    def run():
        dw = torch.full((cout, k, k, cin), 0.5, device="cuda")
        ops.conv2d_wgrad(d, xd, dyd, dw, None)
        torch.cuda.synchronize()
        return dw
    lib = _lib.load()
    _lib.check(lib.yolo_set_wgrad_workspace(None, 0), "yolo_set_wgrad_workspace")     # atomics
    ops._WGRAD_WS = None
    dw_a = run()
    ops.ensure_wgrad_workspace()
    dw1, dw2 = run(), run()
    assert torch.equal(dw1, dw2)
    want = _krsc(wk.grad)
    assert _relerr((dw1 - 0.5).double().cpu(), want) < TOL
    assert _relerr((dw1 - 0.5).double(), (dw_a - 0.5).double()) < 1e-5


def test_conv_rejects_bad_descriptor():
    from tf2_yolo_amd import ops
    from tf2_yolo_amd._lib import YoloHipError
    d = ops.conv_desc((1, 8, 8, 32), 32, 3, 3, 1, "same")
    x = torch.zeros(1, 8, 8, 16, device="cuda")
    w = torch.zeros(32, 3, 3, 32, device="cuda")
    with pytest.raises(YoloHipError):
        ops.conv2d_fwd(d, x, w)


@pytest.mark.parametrize("case", [CASES[0], CASES[2], CASES[4], CASES[5], CASES[8], CASES[13]])
def test_conv_fused_bn_statistics(case):
    """epilogue-fused per-channel sum / sum of squares (training-mode BN) vs the oracle's conv output"""
    from tf2_yolo_amd import ops
    n, h, w, cin, cout, k, s, pad, bias = case
    x, wk, b = _mk(case, seed=5)
    ref = L.conv2d(x, wk, b, stride=s, padding=pad).reshape(-1, cout)
    d = ops.conv_desc((n, h, w, cin), cout, k, k, s, pad)
    stats = torch.zeros(ops.BN_STAT_SLOTS * 2 * cout, device="cuda", dtype=torch.float64)
    amax = torch.zeros(cout, device="cuda", dtype=torch.int32)
    y = ops.conv2d_fwd(d, x.float().cuda(), _krsc(wk).float().cuda(), None if b is None else b.float().cuda(),
                       stats=stats, absmax=amax)
    torch.cuda.synchronize()
    # per-channel max|y| (bit patterns): exactly the maximum of what the kernel wrote
    assert torch.equal(amax.view(torch.float32), y.reshape(-1, cout).abs().max(0).values)
    got = stats.cpu().reshape(ops.BN_STAT_SLOTS, 2, cout).sum(0)
    assert _relerr(got[0], ref.sum(0)) < 1e-5 * max(1.0, (ref.abs().sum(0).max() / ref.sum(0).abs().max()).item())
    assert _relerr(got[1], (ref * ref).sum(0)) < 1e-5


def test_conv_bn_statistics_with_a_large_mean():
    """Channels whose |mean| is 20-80x their standard deviation: a conv bias in front of BatchNormalization
    (yolov{1_5,2}/models/backbone.py, Conv2D's use_bias default). var = E[y^2] - mean^2 from fp32 per-lane partial sums
    would lose ~1e-7 mean^2 / var (measured 2e-4 at these ratios). The product path therefore never
    forms those sums: in training mode the bias cancels in (y - mean), so the executor runs the convolution WITHOUT it and
    hands it to yolo_bn_finalize_offset, where only the moving mean adds it back (engine.forward). Checked here on the
    same calls: the variance to 1e-4 (measured ~1e-7), saved mean / scale / shift of the bias-free tensor, the moving mean
    WITH the bias, and the normalised output equal to the oracle's BatchNormalization of the biased tensor."""
    from tf2_yolo_amd import ops
    case = (4, 52, 52, 32, 128, 3, 1, "same", True)
    n, h, w, cin, cout, k, s_, pad, _ = case
    x, wk, _b = _mk(case, seed=61)
    b = torch.full((cout,), 40.0, dtype=torch.float64) * torch.linspace(0.5, 2.0, cout, dtype=torch.float64)
    ref = L.conv2d(x, wk, b, stride=1, padding=pad)
    r2 = ref.reshape(-1, cout)
    assert float((r2.mean(0).abs() / r2.std(0)).min()) > 15
    d = ops.conv_desc((n, h, w, cin), cout, k, k, 1, pad)
    xd, wd = x.float().cuda(), _krsc(wk).float().cuda()
    xp, wp = ops.split_planes(xd, n * h * w, cin), ops.split_planes(wd, cout, k * k * cin)
    stats = torch.zeros(ops.BN_STAT_SLOTS * 2 * cout, device="cuda", dtype=torch.float64)
    y = ops.conv2d_fwd_planes(d, xp, wp, None, stats=stats)              # no bias: what engine.forward launches
    P = r2.shape[0]
    gamma = (1 + 0.2 * torch.randn(cout, dtype=torch.float64)).float().cuda()
    beta = (0.1 * torch.randn(cout, dtype=torch.float64)).float().cuda()
    mm, mv = torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")
    f = lambda: torch.empty(cout, device="cuda")
    scale, shift, smean, sinv = f(), f(), f(), f()
    ops.bn_finalize(stats, P, cout, gamma, beta, mm, mv, scale, shift, smean, sinv, mean_offset=b.float().cuda())
    torch.cuda.synchronize()
    got = stats.cpu().reshape(ops.BN_STAT_SLOTS, 2, cout).sum(0)
    mean0 = got[0] / P
    var = got[1] / P - mean0 * mean0
    ref_var = r2.var(0, unbiased=False)
    e = ((var - ref_var).abs() / ref_var).max().item()
    print("variance error with the bias left out of the sums:", e)
    assert e < 1e-4
    assert _relerr(mm.double().cpu(), 0.01 * r2.mean(0)) < 1e-6            # moving mean: 0.99 * 0 + 0.01 * (mean + bias)
    assert _relerr(mv.double().cpu(), 0.99 + 0.01 * ref_var) < 1e-6
    z = ops.bn_act_fwd(y, cout, scale, shift, 0)
    zr, _, _ = L.batchnorm_train(ref, gamma.double().cpu(), beta.double().cpu())
    assert _relerr(z.double().cpu(), zr) < TOL


def test_planes_small_magnitude_channels_elementwise():
    """The planes format has ONE power-of-two scale per tensor (csrc/planes.hpp): relative error 2^-22 for elements within
    2^-18 of the bound, absolute error 2^-40 x bound below that. Every other parity test measures max-abs-error over
    max-abs-ref of a whole tensor, which a channel 2^-20 below the tensor's maximum cannot move. This one measures PER
    CHANNEL, relative to that channel's own maximum, through conv -> BatchNormalization (batch statistics) -> LeakyReLU ->
    conv with the input channels spanning 2^0 ... 2^-20 in magnitude: the first convolution is block diagonal (output
    group k reads input group k only), so its output channels span the same 20 octaves; BatchNormalization rescales the
    channels whose variance is above its epsilon (1e-3, Keras' default: the top two groups) to O(1) WITH the relative error
    they had -- below that the reference's own epsilon flattens a channel to beta -- and the second convolution mixes them.
    Bound stated: 1e-4 of each channel's own maximum at every stage, down to 2^-20 of the tensor's maximum (expected from
    the format: 2^-40 x 2 / 2^-20 = 2^-19 = 2e-6 per element at the bottom octave). Reported, not asserted: the 2^-26
    group, where the format is down to ~2^-13."""
    from planes_util import planes_to_dense
    from tf2_yolo_amd import ops
    g = torch.Generator().manual_seed(77)
    n, h, w, cin, cmid, cout, G = 2, 26, 26, 64, 64, 128, 8       # 8 groups of 8 channels
    octaves = [0, -3, -6, -9, -12, -16, -20, -26]
    mag = torch.tensor([2.0 ** o for o in octaves], dtype=torch.float64).repeat_interleave(cin // G)
    x = torch.randn(n, h, w, cin, generator=g, dtype=torch.float64) * mag
    w1 = torch.zeros(3, 3, cin, cmid, dtype=torch.float64)
    for k in range(G):
        sl = slice(k * 8, k * 8 + 8)
        w1[:, :, sl, sl] = torch.randn(3, 3, 8, 8, generator=g, dtype=torch.float64) / (9 * 8) ** 0.5
    w2 = torch.randn(3, 3, cmid, cout, generator=g, dtype=torch.float64) / (9 * cmid) ** 0.5
    gamma = 1 + 0.2 * torch.randn(cmid, generator=g, dtype=torch.float64)
    beta = 0.1 * torch.randn(cmid, generator=g, dtype=torch.float64)
    # float64 oracle
    y1 = L.conv2d(x, w1, None, stride=1, padding="same")
    z1, _, _ = L.batchnorm_train(y1, gamma, beta)
    a1 = L.leaky(z1)
    y2 = L.conv2d(a1, w2, None, stride=1, padding="same")
    # device
    P = n * h * w
    d1 = ops.conv_desc((n, h, w, cin), cmid, 3, 3, 1, "same")
    d2 = ops.conv_desc((n, h, w, cmid), cout, 3, 3, 1, "same")
    xp = ops.split_planes(x.float().cuda(), P, cin)
    w1p = ops.split_planes(_krsc(w1).float().cuda(), cmid, 9 * cin)
    w2p = ops.split_planes(_krsc(w2).float().cuda(), cout, 9 * cmid)
    stats = torch.zeros(ops.BN_STAT_SLOTS * 2 * cmid, device="cuda", dtype=torch.float64)
    y1d = ops.conv2d_fwd_planes(d1, xp, w1p, None, stats=stats)
    f = lambda: torch.empty(cmid, device="cuda")
    scale, shift, smean, sinv = f(), f(), f(), f()
    mm, mv = torch.zeros(cmid, device="cuda"), torch.ones(cmid, device="cuda")
    bnb = torch.zeros(1, device="cuda", dtype=torch.int32)
    ops.bn_finalize(stats, P, cmid, gamma.float().cuda(), beta.float().cuda(), mm, mv, scale, shift, smean, sinv, bound=bnb)
    pl = torch.zeros(ops.planes_bytes(P, cmid), device="cuda", dtype=torch.uint8)
    a1d = ops.bn_act_fwd(y1d, cmid, scale, shift, 1, None, planes=pl, bn_bound=bnb)       # 1 = LeakyReLU(0.1)
    y2d = ops.conv2d_fwd_planes(d2, pl, w2p)
    torch.cuda.synchronize()
    a1_planes, _, _, _ = planes_to_dense(pl.cpu(), P, cmid)     # what the second convolution actually reads

    def per_channel(got, ref):
        got, ref = got.double().cpu().reshape(-1, ref.shape[-1]), ref.reshape(-1, ref.shape[-1])
        return (got - ref).abs().amax(0) / ref.abs().amax(0)
    e_y1 = per_channel(y1d, y1).reshape(G, -1).amax(1)
    e_a1 = per_channel(a1d, a1).reshape(G, -1).amax(1)
    e_pl = per_channel(a1_planes, a1).reshape(G, -1).amax(1)
    e_y2 = per_channel(y2d, y2)
    print("per-channel relative error by input octave", octaves)
    print("  conv1 output      ", [f"{v:.1e}" for v in e_y1.tolist()])
    print("  BN + Leaky (fp32) ", [f"{v:.1e}" for v in e_a1.tolist()])
    print("  BN + Leaky planes ", [f"{v:.1e}" for v in e_pl.tolist()])
    print("  conv2 output (worst channel)", f"{float(e_y2.max()):.1e}")
    ok = [i for i, o in enumerate(octaves) if o >= -20]
    assert float(e_y1[ok].max()) < TOL and float(e_a1[ok].max()) < TOL and float(e_pl[ok].max()) < TOL, (e_y1, e_a1, e_pl)
    # the second convolution against the oracle restricted to the asserted octaves' contribution is not separable: bound the
    # whole output by the asserted stage errors plus the reported bottom group's share (8 of 64 channels at its error)
    assert float(e_y2.max()) < TOL + float(e_pl[-1]) * 8 / 64 * 4, e_y2.max()


# ---- pre-split ("planes") operands + LDS-DMA kernels (include/yolo_hip.h: yolo_split_planes,
# ---- yolo_conv2d_fwd_planes, yolo_conv2d_dgrad_planes) ------------------------------------------------
# (N, H, W, Cin, Cout, k, stride, padding, bias): Cin % 16 == 0 and Cout >= 32
PLANES_CASES = [
    (2, 16, 16, 32, 64, 3, 1, "same", False),        # 128x64 tile, spare loader waves
    (2, 17, 13, 32, 64, 3, 2, "darknet_s2", False),   # stride 2, odd sizes
    (2, 16, 16, 48, 128, 3, 2, "darknet_s2", True),   # Cin % 32 != 0
    (2, 13, 13, 128, 255, 1, 1, "same", True),       # head: Cout = 255 (zero block past the last filter row)
    (2, 7, 7, 96, 160, 3, 1, "same", False),         # M tail (98 pixels), N tail (160 = 128 + 32)
    (3, 9, 11, 256, 128, 1, 1, "valid", False),
    (1, 2, 2, 1024, 512, 3, 1, "same", False),       # 4 pixels: every tap but the centre hits padding
    (2, 14, 14, 64, 64, 3, 2, "same", True),         # 'same' stride 2 (asymmetric pad)
    (1, 13, 13, 64, 125, 1, 1, "same", True),
    (2, 16, 16, 64, 32, 1, 1, "same", False),        # Cout = 32: 128x32 tile (B block loaded by all four waves)
    (2, 15, 17, 32, 64, 3, 2, "darknet_s2", False),   # dgrad with 32 output columns (Cin = 32), 4 parity launches
    # >= 5000 output rows: the epilogue stages the tile in LDS and stores dwordx4 rows (also the accumulate form)
    (2, 51, 53, 32, 160, 3, 1, "same", True),         # 128x128 tiles, partial column tile, partial last row tile
    (2, 51, 53, 160, 32, 1, 1, "same", False),        # 128x32 tile forward; its dgrad: 160 columns
    (3, 60, 60, 64, 64, 3, 2, "darknet_s2", False),   # dgrad: 4 parity classes of 2700 rows (scalar) / fwd 2700 rows
    (2, 72, 72, 48, 64, 3, 1, "same", True),          # 128x64 tile, 10368 rows
    # stride-2 data gradients with >= 128 output columns: the four parity classes in ONE launch under the 128x128 tile
    # (class 0 has a single tap there: must not be mistaken for a 1x1 layer)
    (2, 18, 14, 128, 64, 3, 2, "darknet_s2", False),
    (2, 17, 13, 160, 96, 3, 2, "same", True),         # odd sizes (classes of different tile counts), column tail
]


def test_split_planes_accuracy_and_layout():
    """planes format (csrc/planes.hpp): |s*x - h - l| <= max(2^-22 |s*x|, 2^-25), s a power of two with
    s*max|x| in (2^14, 2^15]; rows past the end and the extra block are zero; header = {bits of max|x|, s, 1/s}"""
    from tf2_yolo_amd import ops
    g = torch.Generator().manual_seed(11)
    rows, c = 37, 48
    x = (torch.randn(rows, c, generator=g) * torch.logspace(-9, 0, rows * c).reshape(rows, c)).float()
    x[0, :3] = torch.tensor([0.0, 1e-30, -2.5])
    pl = ops.split_planes(x.cuda(), rows, c).cpu()
    nblk = (rows + 15) // 16
    body = (nblk + 1) * (c // 16) * 1024
    assert pl.numel() == body + 256 == ops.planes_bytes(rows, c)
    hdr = pl[body:body + 12].view(torch.float32)
    amax = x.abs().max()
    assert hdr[0] == amax and hdr[1] * hdr[2] == 1.0
    s = float(hdr[1])
    assert 2.0 ** 14 < s * float(amax) <= 2.0 ** 15 and s == 2.0 ** round(torch.log2(hdr[1]).item())
    u = pl[:body].view(torch.float16).reshape(nblk + 1, c // 16, 2, 2, 16, 8)       # [blk][kb][plane][half][row][8]
    f = u.permute(2, 0, 4, 1, 3, 5).reshape(2, (nblk + 1) * 16, c).double()         # [plane][row][channel]
    rec = (f[0] + f[1])[:rows]
    t = x.double() * s
    err = (rec - t).abs()
    assert (err <= torch.maximum(2.0 ** -22 * t.abs(), torch.tensor(2.0 ** -25, dtype=torch.float64))).all()
    assert (f[:, rows:] == 0).all()


@pytest.mark.parametrize("case", PLANES_CASES)
def test_conv_fwd_planes(case):
    from tf2_yolo_amd import ops
    n, h, w, cin, cout, k, s, pad, bias = case
    x, wk, b = _mk(case, seed=7)
    ref = L.conv2d(x, wk, b, stride=s, padding=pad)
    d = ops.conv_desc((n, h, w, cin), cout, k, k, s, pad)
    xd, wd = x.float().cuda(), _krsc(wk).float().cuda()
    bd = None if b is None else b.float().cuda()
    xp = ops.split_planes(xd, n * h * w, cin)
    wp = ops.split_planes(wd, cout, k * k * cin)
    stats = torch.zeros(ops.BN_STAT_SLOTS * 2 * cout, device="cuda", dtype=torch.float64)
    amax = torch.zeros(cout, device="cuda", dtype=torch.int32)
    y = ops.conv2d_fwd_planes(d, xp, wp, bd, stats=stats, absmax=amax)
    torch.cuda.synchronize()
    assert torch.equal(amax.view(torch.float32), y.reshape(-1, cout).abs().max(0).values)
    assert _relerr(y.double().cpu(), ref) < TOL
    # the exact bf16 x 6 kernel (register-staged, fp32 operands) and the fp16 x 3 planes kernel agree to a few
    # fp32 roundings of the accumulated magnitude
    if cin % 32 == 0 and ops.CONV_MODE == "split":
        assert _relerr(y.double(), ops.conv2d_fwd(d, xd, wd, bd).double()) < 1e-5
    got = stats.cpu().reshape(ops.BN_STAT_SLOTS, 2, cout).sum(0)
    r2 = ref.reshape(-1, cout)
    assert _relerr(got[1], (r2 * r2).sum(0)) < 1e-5


@pytest.mark.parametrize("case", [c for c in PLANES_CASES if c[4] % 16 == 0 and c[3] >= 32])
def test_conv_dgrad_planes(case):
    from tf2_yolo_amd import ops
    n, h, w, cin, cout, k, s, pad, bias = case
    x, wk, b = _mk(case, seed=8)
    x.requires_grad_(True)
    ref = L.conv2d(x, wk, None, stride=s, padding=pad)
    g = torch.Generator().manual_seed(9)
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(dy)
    d = ops.conv_desc((n, h, w, cin), cout, k, k, s, pad)
    wT = ops.filter_transpose(_krsc(wk.detach()).float().cuda(), cout, k * k, cin)
    dyp = ops.split_planes(dy.float().cuda(), n * d.Ho * d.Wo, cout)
    wTp = ops.split_planes(wT, cin, k * k * cout)
    dx = ops.conv2d_dgrad_planes(d, dyp, wTp)
    torch.cuda.synchronize()
    assert _relerr(dx.double().cpu(), x.grad) < TOL
    dx2 = dx.clone()
    ops.conv2d_dgrad_planes(d, dyp, wTp, dx=dx2, accumulate=True)
    torch.cuda.synchronize()
    assert _relerr(dx2.double().cpu(), 2 * x.grad) < TOL


# ---- 3x3 stride-1 kernel that keeps the input window in LDS (csrc/conv_win.hip; yolo_set_option key 0) ----
WIN_CASES = [
    (2, 7, 7, 96, 160, 3, 1, "same", False),       # M tail (98 pixels), N tail, 6 channel blocks
    (1, 2, 2, 1024, 512, 3, 1, "same", False),     # 4 pixels, 64 channel blocks
    (3, 13, 13, 32, 128, 3, 1, "same", True),      # tiles cross image boundaries (169 pixels per image)
    (2, 26, 26, 48, 128, 3, 1, "same", False),     # odd number of channel blocks (3)
    (2, 51, 53, 16, 256, 3, 1, "same", True),      # one channel block, two column tiles, odd row length
    (5, 5, 9, 64, 128, 3, 1, "same", False),       # tiny images: a tile spans three of them
    (1, 104, 104, 32, 128, 3, 1, "same", False),   # long rows: wide window
    (2, 20, 17, 128, 256, 3, 1, "same", False),    # data gradient with 128 columns: window kernel both ways
]


@pytest.mark.parametrize("sk", [0, 1, -1, 5])
@pytest.mark.parametrize("win", [2, 4])
@pytest.mark.parametrize("case", WIN_CASES)
def test_conv_window_kernel_fwd_dgrad(case, win, sk):
    """sk = 0: one workgroup per tile; 1: split-K + reduce kernel (launches without statistics: the per-tap y0 and the
    data gradients here); -1: the stream-K policy; 5: stream-K forced onto five workgroups (parts of a tile combined by
    the last arriver, workgroups spanning tile boundaries)"""
    from tf2_yolo_amd import ops
    ops.ensure_conv_workspace()
    ops.set_option(ops.OPT_CONV_SK, sk)
    n, h, w, cin, cout, k, s, pad, bias = case
    x, wk, b = _mk(case, seed=21)
    x.requires_grad_(True)
    ref = L.conv2d(x, wk, b, stride=s, padding=pad)
    g = torch.Generator().manual_seed(22)
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(dy)
    d = ops.conv_desc((n, h, w, cin), cout, k, k, s, pad)
    xd, wd = x.detach().float().cuda(), _krsc(wk).float().cuda()
    bd = None if b is None else b.float().cuda()
    xp = ops.split_planes(xd, n * h * w, cin)
    wp = ops.split_planes(wd, cout, k * k * cin)
    stats = torch.zeros(ops.BN_STAT_SLOTS * 2 * cout, device="cuda", dtype=torch.float64)
    amax = torch.zeros(cout, device="cuda", dtype=torch.int32)
    ops.set_option(ops.OPT_CONV_WIN, 0)
    y0 = ops.conv2d_fwd_planes(d, xp, wp, bd)
    ops.set_option(ops.OPT_CONV_WIN, win)
    try:
        y = ops.conv2d_fwd_planes(d, xp, wp, bd, stats=stats, absmax=amax)
        torch.cuda.synchronize()
        assert _relerr(y.double().cpu(), ref.detach()) < TOL
        assert _relerr(y.double(), y0.double()) < 1e-5      # same arithmetic as the per-tap kernel, other summation order
        assert torch.equal(amax.view(torch.float32), y.reshape(-1, cout).abs().max(0).values)
        got = stats.cpu().reshape(ops.BN_STAT_SLOTS, 2, cout).sum(0)
        r2 = ref.detach().reshape(-1, cout)
        assert _relerr(got[1], (r2 * r2).sum(0)) < 1e-5
        if cin >= 32:
            # data gradient = the same kernel on dy with the transposed filter (needs Cin >= 128 columns for the
            # window kernel; narrower ones fall through to the per-tap kernel and must still be right)
            wT = ops.filter_transpose(wd, cout, k * k, cin)
            dyp = ops.split_planes(dy.float().cuda(), n * d.Ho * d.Wo, cout)
            wTp = ops.split_planes(wT, cin, k * k * cout)
            dx = ops.conv2d_dgrad_planes(d, dyp, wTp)
            torch.cuda.synchronize()
            assert _relerr(dx.double().cpu(), x.grad) < TOL
            dx2 = dx.clone()
            ops.conv2d_dgrad_planes(d, dyp, wTp, dx=dx2, accumulate=True)
            torch.cuda.synchronize()
            assert _relerr(dx2.double().cpu(), 2 * x.grad) < TOL
    finally:
        ops.reset_options()


# ---- window kernel on 2-D patches (conv_win.hip GEO = 1): rows longer than 64 pixels, 64-column tiles ----
PATCH_CASES = [
    (2, 76, 76, 32, 128, 3, 1, "same", False),     # YOLOv4-608 stage 3 rows: 9.5 x 4.75 patches per image (ragged both ways)
    (1, 104, 104, 64, 128, 3, 1, "same", True),    # YOLOv3-416 block 2 rows, bias
    (1, 152, 152, 32, 64, 3, 1, "same", False),    # Cout = 64: 256 x 64 tiles on 16 x 16 patches (9.5 x 9.5 per image)
    (2, 208, 24, 32, 64, 3, 1, "same", True),      # 208-pixel columns, 1.5 patches across; bias; Cin = 32 (two channel blocks)
    (3, 9, 21, 48, 160, 3, 1, "same", False),      # images smaller than a patch row count, ragged column tile (160 = 128 + 32)
    (1, 37, 130, 16, 64, 3, 1, "same", False),     # ONE channel block (9 stages), odd sizes
    (2, 40, 72, 128, 64, 3, 1, "same", False),     # data gradient with 128 columns (128 x 128 tiles), forward with 64
]


@pytest.mark.parametrize("case", PATCH_CASES)
def test_conv_patch_window_kernel_fwd_dgrad(case):
    """conv_win_kernel<..., GEO = 1>: the tile is an 8 x 16 (Cout > 64) or 16 x 16 (Cout <= 64) patch of output pixels,
    the window its halo patch. Forced with yolo_set_option(5, 2) so that the small cases reach it; against the float64
    oracle (1e-4) and the per-tap kernel (1e-5): forward with BatchNorm statistics and per-channel max|y| (rows of a ragged
    patch that lie outside the image must not count), data gradient plain and accumulating."""
    from tf2_yolo_amd import ops
    ops.ensure_conv_workspace()
    n, h, w, cin, cout, k, s, pad, bias = case
    x, wk, b = _mk(case, seed=31)
    x.requires_grad_(True)
    ref = L.conv2d(x, wk, b, stride=s, padding=pad)
    g = torch.Generator().manual_seed(32)
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(dy)
    d = ops.conv_desc((n, h, w, cin), cout, k, k, s, pad)
    xd, wd = x.detach().float().cuda(), _krsc(wk).float().cuda()
    bd = None if b is None else b.float().cuda()
    xp = ops.split_planes(xd, n * h * w, cin)
    wp = ops.split_planes(wd, cout, k * k * cin)
    stats = torch.zeros(ops.BN_STAT_SLOTS * 2 * cout, device="cuda", dtype=torch.float64)
    amax = torch.zeros(cout, device="cuda", dtype=torch.int32)
    ops.set_option(ops.OPT_CONV_WIN, 0)
    y0 = ops.conv2d_fwd_planes(d, xp, wp, bd)
    ops.set_option(ops.OPT_CONV_WIN, 1)
    ops.set_option(ops.OPT_CONV_PATCH, 2)
    try:
        y = ops.conv2d_fwd_planes(d, xp, wp, bd, stats=stats, absmax=amax)
        torch.cuda.synchronize()
        assert _relerr(y.double().cpu(), ref.detach()) < TOL
        assert _relerr(y.double(), y0.double()) < 1e-5
        assert torch.equal(amax.view(torch.float32), y.reshape(-1, cout).abs().max(0).values)
        got = stats.cpu().reshape(ops.BN_STAT_SLOTS, 2, cout).sum(0)
        r2 = ref.detach().reshape(-1, cout)
        assert _relerr(got[0], r2.sum(0)) < 1e-5 and _relerr(got[1], (r2 * r2).sum(0)) < 1e-5
        # no statistics asked for (inference / data gradients take this path through the epilogue)
        y1 = ops.conv2d_fwd_planes(d, xp, wp, bd)
        assert torch.equal(y1, y)
        if cin >= 64:
            wT = ops.filter_transpose(wd, cout, k * k, cin)
            dyp = ops.split_planes(dy.float().cuda(), n * d.Ho * d.Wo, cout)
            wTp = ops.split_planes(wT, cin, k * k * cout)
            dx = ops.conv2d_dgrad_planes(d, dyp, wTp)
            torch.cuda.synchronize()
            assert _relerr(dx.double().cpu(), x.grad) < TOL
            dx2 = dx.clone()
            ops.conv2d_dgrad_planes(d, dyp, wTp, dx=dx2, accumulate=True)
            torch.cuda.synchronize()
            assert _relerr(dx2.double().cpu(), 2 * x.grad) < TOL
    finally:
        ops.reset_options()


# ---- split-K of launches that leave most of the chip idle (bs-1 inference; conv_win.hip: conv_split_reduce_kernel) ----
@pytest.mark.parametrize("case", [
    (1, 13, 13, 512, 1024, 3, 1, "same", False),       # window kernel: 16 tiles x 16 parts
    (1, 13, 13, 1024, 512, 1, 1, "same", True),        # per-tap kernel, 1x1: 8 tiles x 8 parts; bias
    (1, 26, 26, 256, 512, 3, 1, "same", False),        # window kernel: 24 tiles x 8 parts
    (2, 17, 13, 64, 128, 3, 2, "darknet_s2", False),   # per-tap kernel, stride 2: 4 parts of one channel block
    (1, 19, 19, 80, 160, 3, 1, "same", True),          # 5 channel blocks (uneven parts), ragged column tile
])
def test_conv_split_k(case):
    """YOLO_CONV_SK=1 (the default): every tile computed by several workgroups, the reduce kernel adds the parts in
    order and runs the epilogue. Forward (plain, and with the fused inference epilogue + residual + per-channel max),
    data gradient (plain and accumulating) against the float64 oracle at 1e-4 and against the unsplit launch at 1e-5;
    two runs are bit-identical (ordered sum, no atomics on the values)."""
    from tf2_yolo_amd import ops
    from tf2_yolo_amd._lib import ACT_LEAKY
    ops.ensure_conv_workspace()
    n, h, w, cin, cout, k, s, pad, bias = case
    x, wk, b = _mk(case, seed=71)
    x.requires_grad_(True)
    ref = L.conv2d(x, wk, b, stride=s, padding=pad)
    g = torch.Generator().manual_seed(72)
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(dy)
    d = ops.conv_desc((n, h, w, cin), cout, k, k, s, pad)
    xd, wd = x.detach().float().cuda(), _krsc(wk).float().cuda()
    bd = None if b is None else b.float().cuda()
    xp, wp = ops.split_planes(xd, n * h * w, cin), ops.split_planes(wd, cout, k * k * cin)
    scale = (1 + 0.2 * torch.randn(cout, generator=g)).cuda()
    shift = (0.1 * torch.randn(cout, generator=g)).cuda()
    res = torch.randn(ref.shape, generator=g).cuda()
    wT = ops.filter_transpose(wd, cout, k * k, cin)
    dyp = ops.split_planes(dy.float().cuda(), n * d.Ho * d.Wo, cout)
    wTp = ops.split_planes(wT, cin, k * k * cout)

    def run():
        amax = torch.zeros(cout, device="cuda", dtype=torch.int32)
        y = ops.conv2d_fwd_planes(d, xp, wp, bd)
        ye = ops.conv2d_fwd_planes_epi(d, xp, wp, bd, ops.EPI_AFFINE_LEAKY, scale, shift, residual=res, absmax=amax)
        dx = ops.conv2d_dgrad_planes(d, dyp, wTp)
        dx2 = dx.clone()
        ops.conv2d_dgrad_planes(d, dyp, wTp, dx=dx2, accumulate=True)
        torch.cuda.synchronize()
        return y, ye, dx, dx2, amax

    try:
        ops.set_option(ops.OPT_CONV_SK, 0)
        y0, ye0, dx0, dx20, amax0 = run()
        ops.set_option(ops.OPT_CONV_SK, 1)
        y, ye, dx, dx2, amax = run()
        again = run()
    finally:
        ops.reset_options()
    assert not torch.equal(y, y0) or not torch.equal(dx, dx0), "the split-K path did not run"
    assert _relerr(y.double().cpu(), ref.detach()) < TOL and _relerr(dx.double().cpu(), x.grad) < TOL
    assert _relerr(dx2.double().cpu(), 2 * x.grad) < TOL
    z = L.leaky(ref.detach() * scale.double().cpu() + shift.double().cpu())
    assert _relerr(ye.double().cpu(), z + res.double().cpu()) < TOL
    for a, a0 in ((y, y0), (ye, ye0), (dx, dx0), (dx2, dx20)):
        assert _relerr(a.double(), a0.double()) < 1e-5
    # per-channel max|.| before the residual
    pre = ops.bn_act_fwd(y, cout, scale, shift, ACT_LEAKY)
    assert torch.equal(amax.view(torch.float32), pre.reshape(-1, cout).abs().max(0).values)
    for a, a1 in zip((y, ye, dx, dx2, amax), again):
        assert torch.equal(a, a1)


# ---- the benchmark's own layer sizes against the float64 oracle (not against another device kernel) ----
# YOLOv3-416 at bs 32, coarsest level: M = 5408 rows, K = 4608 (3x3, 512 -> 1024) and 1024 (1x1, 1024 -> 512)
@pytest.mark.parametrize("case", [(32, 13, 13, 512, 1024, 3, 1, "same", False), (32, 13, 13, 1024, 512, 1, 1, "same", False)])
def test_conv_planes_c3_layer_sizes_vs_fp64_oracle(case):
    from tf2_yolo_amd import ops
    n, h, w, cin, cout, k, s, pad, bias = case
    x, wk, _ = _mk(case, seed=31)
    x.requires_grad_(True)
    wk.requires_grad_(True)
    ref = L.conv2d(x, wk, None, stride=s, padding=pad)
    g = torch.Generator().manual_seed(32)
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(dy)
    d = ops.conv_desc((n, h, w, cin), cout, k, k, s, pad)
    xd, wd, dyd = x.detach().float().cuda(), _krsc(wk.detach()).float().cuda(), dy.float().cuda()
    xp, wp = ops.split_planes(xd, n * h * w, cin), ops.split_planes(wd, cout, k * k * cin)
    dyp = ops.split_planes(dyd, n * h * w, cout)
    wTp = ops.split_planes(ops.filter_transpose(wd, cout, k * k, cin), cin, k * k * cout)
    y = ops.conv2d_fwd_planes(d, xp, wp)
    dx = ops.conv2d_dgrad_planes(d, dyp, wTp)
    dw = torch.zeros(cout, k, k, cin, device="cuda")
    ops.conv2d_wgrad_planes(d, xp, dyp, dw)
    torch.cuda.synchronize()
    e = (_relerr(y.double().cpu(), ref.detach()), _relerr(dx.double().cpu(), x.grad), _relerr(dw.double().cpu(), _krsc(wk.grad)))
    print("C3 layer", case[:7], "fwd / dgrad / wgrad error vs fp64:", e)
    assert max(e) < 2e-5, e      # measured 2e-6 .. 4e-6: fifty times inside the 1e-4 parity bar


def _planes_with_bound(x2d, bound):
    """planes of x2d [P][C] under an explicit upper bound of max|x| (yolo_bn_act_fwd_planes as an identity)"""
    from tf2_yolo_amd import ops
    from tf2_yolo_amd._lib import ACT_LINEAR
    P, C = x2d.shape
    pl = torch.zeros(ops.planes_bytes(P, C), dtype=torch.uint8, device="cuda")
    b = torch.tensor([bound], dtype=torch.float32, device="cuda").view(torch.int32)
    ops.bn_act_fwd(x2d, C, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda"), ACT_LINEAR, planes=pl,
                   bn_bound=b, want_out=False)
    return pl


@pytest.mark.parametrize("mode", ["loose_bound", "outlier"])
def test_conv_planes_scaling_stress(mode):
    """The planes carry ONE power-of-two scale per tensor, chosen from an upper bound of max|x|: small elements of a
    tensor with a loose bound or a huge outlier fall into fp16's subnormal range (absolute error 2^-40 x bound).
    loose_bound: the bound bn_finalize falls back to without per-channel maxima, sqrt(P) times the true maximum.
    outlier: one element 10^6 times the rest; the error is measured on the outputs it does not touch."""
    from tf2_yolo_amd import ops
    n, h, w, cin, cout, k = 2, 51, 53, 128, 128, 3
    g = torch.Generator().manual_seed(41)
    x = torch.randn(n, h, w, cin, generator=g, dtype=torch.float64)
    wk = torch.randn(k, k, cin, cout, generator=g, dtype=torch.float64) / (k * k * cin) ** 0.5
    P = n * h * w
    touched = torch.zeros(n, h, w, dtype=torch.bool)
    if mode == "outlier":
        x[1, 20, 30, 7] = 1.0e6
        touched[1, 19:22, 29:32] = True
        bound = float(x.abs().max())
    else:
        bound = float(x.abs().max()) * P ** 0.5
    ref = L.conv2d(x, wk, None, stride=1, padding="same")
    d = ops.conv_desc((n, h, w, cin), cout, k, k, 1, "same")
    xp = _planes_with_bound(x.float().cuda().reshape(P, cin), bound)
    wp = ops.split_planes(_krsc(wk).float().cuda(), cout, k * k * cin)
    y = ops.conv2d_fwd_planes(d, xp, wp).double().cpu()
    torch.cuda.synchronize()
    keep = ~touched
    err = (y - ref)[keep].abs().max().item() / ref[keep].abs().max().item()
    print("scaling stress", mode, "bound / max|x| =", bound / float(x[keep].abs().max()), "error", err)
    assert err < TOL, err
    if mode == "outlier":   # the touched outputs: relative to THEIR scale
        assert (y - ref)[touched].abs().max().item() / ref[touched].abs().max().item() < TOL


@pytest.mark.parametrize("act,with_res,case", [
    ("leaky", False, (2, 13, 13, 64, 128, 3, 1, "same", False)),
    ("leaky", True, (2, 26, 26, 128, 128, 3, 1, "same", False)),       # residual block: x + act(BN(conv(.)))
    ("mish", True, (1, 19, 19, 64, 64, 3, 1, "same", False)),          # v4 CSP block
    ("mish", False, (2, 16, 16, 32, 64, 1, 1, "same", True)),          # 1x1, conv bias in front of BN (v2 style)
    ("leaky", False, (2, 17, 13, 32, 64, 3, 2, "darknet_s2", False)),  # down-sampling conv
])
def test_conv_fused_inference_epilogue(act, with_res, case):
    """yolo_conv2d_fwd_planes_epi (affine+leaky / affine+mish epilogue): against the float64
    oracle (conv2d -> batchnorm_infer -> leaky | mish [-> + residual]) at 1e-4, and BIT-identical to the unfused device
    path (conv kernel, then yolo_bn_act_fwd) -- same fp32 operations in the same order; then yolo_split_planes_absmax:
    the planes it produces feed a second convolution that must again match the oracle."""
    from tf2_yolo_amd import ops
    from tf2_yolo_amd._lib import ACT_LEAKY, ACT_MISH
    n, h, w, cin, cout, k, s, pad, bias = case
    x, wk, b = _mk(case, seed=51)
    g = torch.Generator().manual_seed(52)
    gamma = 1 + 0.2 * torch.randn(cout, generator=g, dtype=torch.float64)
    beta = 0.1 * torch.randn(cout, generator=g, dtype=torch.float64)
    mm = 0.1 * torch.randn(cout, generator=g, dtype=torch.float64)
    mv = 0.5 + torch.rand(cout, generator=g, dtype=torch.float64)
    yc = L.conv2d(x, wk, b, stride=s, padding=pad)
    res = torch.randn(yc.shape, generator=g, dtype=torch.float64) if with_res else None
    z = L.batchnorm_infer(yc, gamma, beta, mm, mv)
    ref = (L.leaky(z) if act == "leaky" else L.mish(z)) + (res if with_res else 0)
    d = ops.conv_desc((n, h, w, cin), cout, k, k, s, pad)
    xd, wd = x.float().cuda(), _krsc(wk).float().cuda()
    bd = None if b is None else b.float().cuda()
    scale, shift = torch.empty(cout, device="cuda"), torch.empty(cout, device="cuda")
    ops.bn_fold_inference(cout, gamma.float().cuda(), beta.float().cuda(), mm.float().cuda(), mv.float().cuda(), scale, shift)
    xp, wp = ops.split_planes(xd, n * h * w, cin), ops.split_planes(wd, cout, k * k * cin)
    resd = res.float().cuda() if with_res else None
    amax = torch.zeros(cout, device="cuda", dtype=torch.int32)
    epi = ops.EPI_AFFINE_LEAKY if act == "leaky" else ops.EPI_AFFINE_MISH
    y = ops.conv2d_fwd_planes_epi(d, xp, wp, bd, epi, scale, shift, residual=resd, absmax=amax)
    torch.cuda.synchronize()
    assert _relerr(y.double().cpu(), ref) < TOL
    # unfused device path
    y0 = ops.conv2d_fwd_planes(d, xp, wp, bd)
    a0 = ops.bn_act_fwd(y0, cout, scale, shift, ACT_LEAKY if act == "leaky" else ACT_MISH, residual=resd)
    torch.cuda.synchronize()
    assert torch.equal(y, a0)
    # absmax = per-channel max of |act(BN(conv))| BEFORE the residual
    pre = ops.bn_act_fwd(y0, cout, scale, shift, ACT_LEAKY if act == "leaky" else ACT_MISH)
    assert torch.equal(amax.view(torch.float32), pre.reshape(-1, cout).abs().max(0).values)
    # the planes of y for the next conv (bound = max of absmax (+ bound of the residual)); a 1x1 conv on them vs the oracle
    rows = n * d.Ho * d.Wo
    pl = torch.zeros(ops.planes_bytes(rows, cout), dtype=torch.uint8, device="cuda")
    rb = torch.tensor([float(res.abs().max())], device="cuda") if with_res else None
    ob = torch.zeros(1, device="cuda")
    ops.split_planes_absmax(y, rows, cout, amax, pl, extra_bound=rb, out_bound=ob)
    assert float(ob.item()) >= float(y.abs().max()) > 0
    w2 = torch.randn(1, 1, cout, 64, generator=g, dtype=torch.float64) / cout ** 0.5
    d2 = ops.conv_desc((n, d.Ho, d.Wo, cout), 64, 1, 1, 1, "same")
    y2 = ops.conv2d_fwd_planes(d2, pl, ops.split_planes(_krsc(w2).float().cuda(), 64, cout))
    torch.cuda.synchronize()
    assert _relerr(y2.double().cpu(), L.conv2d(ref, w2, None, stride=1, padding="same")) < TOL


def _planes_values(pl, rows, c):
    """the fp64 values a planes buffer holds ((h + l) / scale) and its header (bound, scale)"""
    nblk = (rows + 15) // 16
    body = (nblk + 1) * (c // 16) * 1024
    hdr = pl[body:body + 12].view(torch.float32)
    u = pl[:body].view(torch.float16).reshape(nblk + 1, c // 16, 2, 2, 16, 8)
    f = u.permute(2, 0, 4, 1, 3, 5).reshape(2, (nblk + 1) * 16, c).double()
    return (f[0] + f[1])[:rows] / float(hdr[1]), float(hdr[0]), float(hdr[1]), f[:, rows:]


def _conv_small_takes(rows, cout, k, cin):
    """the launch policy of csrc/conv_small.hip (small_tile), environment overrides included"""
    on = int(os.environ.get("YOLO_CONV_SMALL", "1"))
    if on == 0 or cout % 32 != 0 or k not in (1, 3):
        return False
    g11 = ((rows + 31) // 32) * (cout // 32)
    g22 = ((rows + 63) // 64) * (cout // 64) if cout % 64 == 0 else 1 << 40
    grid = int(os.environ.get("YOLO_CONV_SMALL_GRID", "0"))
    if os.environ.get("YOLO_CONV_SMALL_TILE"):
        return g11 <= min(grid or 2048, 4096)
    if k == 3 and on != 3:
        cap, steps = (grid or 256), 9 * (cin // 16)
        return g22 >= 128 and ((g22 <= cap and steps <= 72) or (g22 <= 2 * cap and steps <= 36))
    return g11 <= (grid or 256) or g22 <= (grid or 256)


@pytest.mark.parametrize("case,with_res,expect_onepass", [
    ((1, 13, 13, 512, 1024, 3, 1, "same", False), True, True),    # window kernel, split-K: the reduce kernel writes the planes
    ((1, 26, 26, 512, 256, 1, 1, "same", False), False, True),   # per-tap kernel, split-K
    ((1, 52, 52, 128, 256, 3, 1, "same", True), True, True),     # conv bias in front of the folded BatchNorm
    ((1, 19, 19, 128, 96, 3, 1, "same", False), True, True),     # rows not a multiple of 16, column tail (96 = 64 + 32)
    ((16, 52, 52, 64, 128, 3, 1, "same", False), True, False),   # tiles fill the chip: two passes inside the call
    # round 6, conv_small.hip (one launch, K split across the waves of a workgroup) -- the cases above with Cout % 32 == 0 too:
    ((1, 13, 13, 1024, 512, 1, 1, "same", False), False, True),  # 64 steps: every wave eight, two register buffers
    ((1, 26, 26, 256, 512, 3, 2, "same", True), False, None),    # stride 2 (Keras 'same': pad 0 / 1), bias (3x3: YOLO_CONV_SMALL=3)
    ((1, 52, 52, 256, 128, 1, 1, "same", False), True, True),    # 2704 pixels: 84.5 row tiles, residual
    ((2, 13, 13, 64, 64, 3, 1, "same", False), True, None),      # two images: taps must not cross from one into the next
    ((1, 40, 40, 32, 64, 3, 1, "same", False), False, None),     # 18 steps on eight waves: ragged
    ((1, 20, 20, 16, 32, 1, 1, "same", False), True, None),      # ONE step: seven waves multiply by the zero block
    ((1, 26, 26, 128, 64, 1, 2, "same", False), False, None),    # 1x1 stride 2
    ((1, 104, 104, 64, 128, 3, 1, "same", False), True, None),   # 338 workgroups of 64 x 64: two rounds
])
def test_inference_unit_writes_its_planes(case, with_res, expect_onepass):
    """yolo_conv2d_fwd_infer_unit: y bit-identical to the fused-epilogue convolution, its planes (scaled by the a-priori
    bound K max|x| + D + max|residual|, yolo_conv_pred_bound) hold y to the format's 22 bits, the recorded bound of the
    result is max|y| (one pass: one word per workgroup) or an upper bound of it (two passes inside the call)"""
    from tf2_yolo_amd import ops
    ops.ensure_conv_workspace()   # (split-K needs its slabs: the engine registers them when a network is built)
    n, h, w, cin, cout, k, s, pad, bias = case
    x, wk, b = _mk(case, seed=21)
    g = torch.Generator().manual_seed(22)
    d = ops.conv_desc((n, h, w, cin), cout, k, k, s, pad)
    rows = n * d.Ho * d.Wo
    xd, wd = x.float().cuda(), _krsc(wk).float().cuda()
    bd = None if b is None else b.float().cuda()
    scale = (torch.rand(cout, generator=g) + 0.5).float().cuda() * torch.where(torch.rand(cout, generator=g) < 0.2, -1.0, 1.0).cuda()
    shift = torch.randn(cout, generator=g).float().cuda() * 0.3
    res = torch.randn(n, d.Ho, d.Wo, cout, generator=g).float().cuda() * 2.0 if with_res else None
    xp = ops.split_planes(xd, n * h * w, cin)
    wp = ops.split_planes(wd, cout, k * k * cin)
    amax0 = torch.zeros(cout, device="cuda", dtype=torch.int32)
    y_ref = ops.conv2d_fwd_planes_epi(d, xp, wp, bd, ops.EPI_AFFINE_LEAKY, scale, shift, residual=res, absmax=amax0)
    in_bound = xd.abs().max().reshape(1)
    res_bound = res.abs().max().reshape(1) if with_res else None
    pred = torch.zeros(2, device="cuda")
    ops.conv_pred_bound(wd, cout, k * k * cin, scale, shift, bd, pred)
    l1 = wd.double().abs().reshape(cout, -1).sum(1)
    K_ref = float((scale.double().abs() * l1).max())
    D_ref = float((scale.double() * (bd.double() if bd is not None else 0.0) + shift.double()).abs().max())
    assert K_ref <= float(pred[0]) <= K_ref * (1 + 1e-5) and D_ref <= float(pred[1]) <= D_ref * (1 + 1e-5) + 1e-30
    amax = torch.zeros(cout, device="cuda", dtype=torch.int32)
    y = torch.empty_like(y_ref)
    pl = torch.zeros(ops.planes_bytes(rows, cout), device="cuda", dtype=torch.uint8)
    out_bound = torch.zeros(1, device="cuda")
    out_words = torch.full((ops.INFER_BOUND_WORDS,), -1, device="cuda", dtype=torch.int32)   # (not zeroed: plain stores)
    # the input's bound as the words another one-pass unit leaves (any number up to 4096), the residual's as one float
    in_words = torch.zeros(777, device="cuda")
    in_words[torch.randint(0, 777, (40,), generator=g)] = in_bound * 0.5
    in_words[500] = in_bound
    nw = ops.conv2d_fwd_infer_unit(d, xp, wp, bd, ops.EPI_AFFINE_LEAKY, scale, shift, res, y, amax, pred,
                                   in_words.view(torch.int32), res_bound, pl, out_words, out_bound)
    torch.cuda.synchronize()
    assert expect_onepass is None or (nw > 0) == expect_onepass
    small = _conv_small_takes(rows, cout, k, cin)   # (csrc/conv_small.hip: small_tile)
    assert not small or nw > 0
    if k == 1 and cout % 32 == 0 and rows <= 2704 and not os.environ.get("YOLO_CONV_SMALL"):
        assert small   # the 1x1 cases of this list are what the kernel was written for
    if small:   # conv_small.hip adds in another order than the kernel behind conv2d_fwd_planes_epi
        assert _relerr(y.double(), y_ref.double()) < 2e-6
    else:
        assert torch.equal(y, y_ref)
    ymax = float(y.abs().max())
    vals, bound, sc, tail = _planes_values(pl.cpu(), rows, cout)
    assert bound >= ymax and 2.0 ** 14 < sc * bound <= 2.0 ** 15
    assert (tail == 0).all()
    yd = y.double().cpu().reshape(rows, cout)
    err = (vals - yd).abs()
    assert (err <= torch.maximum(2.0 ** -22 * yd.abs(), torch.tensor(2.0 ** -25 / sc, dtype=torch.float64))).all()
    if nw:
        assert nw <= ops.INFER_BOUND_WORDS and float(out_words[:nw].view(torch.float32).max()) == ymax
        assert bool((out_words[nw:] == -1).all()) and not bool(amax.any())
        folded = torch.zeros(1, device="cuda")
        ops.fold_bound(out_words[:nw], folded)
        assert float(folded) == ymax
    else:
        ob = float(out_bound)
        assert torch.equal(amax, amax0)
        assert ymax <= ob <= (ymax + (float(res_bound) if with_res else 0.0)) * 1.002 + 1e-30
    # the consumer's view: a 1x1 convolution on these planes against the same convolution on planes split from y
    w2 = (torch.randn(32, cout, generator=g) * 0.05).float().cuda()
    w2p = ops.split_planes(w2, 32, cout)
    d2 = ops.conv_desc((n, d.Ho, d.Wo, cout), 32, 1, 1, 1, "same")
    z1 = ops.conv2d_fwd_planes(d2, pl, w2p)
    z2 = ops.conv2d_fwd_planes(d2, ops.split_planes(y, rows, cout), w2p)
    assert _relerr(z1.double(), z2.double()) < 2e-6


def _small_fuzz_cases():
    import random
    rnd = random.Random(20261005)
    cases = []
    while len(cases) < 24:
        k = rnd.choice([1, 1, 3])
        s_ = rnd.choice([1, 1, 2])
        n = rnd.choice([1, 1, 2, 3])
        h, w = rnd.randint(5, 30), rnd.randint(5, 30)
        cin = 16 * rnd.randint(1, 12)
        cout = 32 * rnd.randint(1, 8)
        pad = rnd.choice(["same", "same", "valid"])
        if pad == "valid" and (h < k + 1 or w < k + 1):
            continue
        cases.append((n, h, w, cin, cout, k, s_, pad, rnd.random() < 0.3))
    return cases


@pytest.mark.parametrize("case", _small_fuzz_cases())
def test_conv_small_kernel_random_shapes(case):
    """yolo_conv2d_fwd_infer_unit on 24 random small shapes (1x1 and 3x3, strides 1 / 2, 'same' / 'valid', 1-3 images, odd
    sizes, Cin = 16 .. 192, Cout = 32 .. 256, with and without a residual): whichever kernel the policy picks
    (csrc/conv_small.hip for the 1x1 units and, under YOLO_CONV_SMALL=3 -- scripts/gpu/r6aw.sh runs the file that way too --
    for every 3x3 unit) against the float64 oracle and the fused-epilogue convolution of the training kernels."""
    from tf2_yolo_amd import ops
    ops.ensure_conv_workspace()
    n, h, w, cin, cout, k, s, pad, bias = case
    x, wk, b = _mk(case, seed=51)
    g = torch.Generator().manual_seed(52)
    d = ops.conv_desc((n, h, w, cin), cout, k, k, s, pad)
    rows = n * d.Ho * d.Wo
    xd, wd = x.float().cuda(), _krsc(wk).float().cuda()
    bd = None if b is None else b.float().cuda()
    scale = (torch.rand(cout, generator=g) + 0.5).float().cuda()
    shift = (torch.randn(cout, generator=g) * 0.3).float().cuda()
    with_res = (h + w) % 2 == 0
    res = torch.randn(n, d.Ho, d.Wo, cout, generator=g).float().cuda() if with_res else None
    xp = ops.split_planes(xd, n * h * w, cin)
    wp = ops.split_planes(wd, cout, k * k * cin)
    amax0 = torch.zeros(cout, device="cuda", dtype=torch.int32)
    y_ref = ops.conv2d_fwd_planes_epi(d, xp, wp, bd, ops.EPI_AFFINE_LEAKY, scale, shift, residual=res, absmax=amax0)
    pred = torch.zeros(2, device="cuda")
    ops.conv_pred_bound(wd, cout, k * k * cin, scale, shift, bd, pred)
    y = torch.full_like(y_ref, float("nan"))
    pl = torch.zeros(ops.planes_bytes(rows, cout), device="cuda", dtype=torch.uint8)
    out_words = torch.full((ops.INFER_BOUND_WORDS,), -1, device="cuda", dtype=torch.int32)
    nw = ops.conv2d_fwd_infer_unit(d, xp, wp, bd, ops.EPI_AFFINE_LEAKY, scale, shift, res, y,
                                   torch.zeros(cout, device="cuda", dtype=torch.int32), pred, xd.abs().max().reshape(1),
                                   res.abs().max().reshape(1) if with_res else None, pl, out_words, torch.zeros(1, device="cuda"))
    torch.cuda.synchronize()
    assert _relerr(y.double(), y_ref.double()) < 2e-6
    ref = L.conv2d(x, wk, b, stride=s, padding=pad)
    z = ref.permute(0, 1, 2, 3) * scale.double().cpu() + shift.double().cpu()
    z = torch.where(z > 0, z, 0.1 * z) + (res.double().cpu() if with_res else 0.0)
    assert _relerr(y.double().cpu(), z) < 1e-5
    vals, bound, sc, tail = _planes_values(pl.cpu(), rows, cout)
    assert (tail == 0).all() and bound >= float(y.abs().max())
    yd = y.double().cpu().reshape(rows, cout)
    assert ((vals - yd).abs() <= torch.maximum(2.0 ** -22 * yd.abs(), torch.tensor(2.0 ** -25 / sc, dtype=torch.float64))).all()
    if nw:
        assert float(out_words[:nw].view(torch.float32).max()) == float(y.abs().max())


@pytest.mark.parametrize("n,hw,cin,A,C,version", [(1, 13, 1024, 3, 80, 3), (1, 26, 512, 3, 80, 3), (1, 52, 256, 3, 80, 4),
                                                  (2, 19, 128, 3, 20, 3), (1, 13, 1024, 5, 20, 2), (8, 52, 256, 3, 80, 3)])
def test_head_unit_in_one_call(n, hw, cin, A, C, version):
    """yolo_conv2d_fwd_head_unit (round 6): the head's 1x1 convolution (bias, Cout = A (5 + C): 255, 75, 125 -- not a
    multiple of 16) and its activation; at few output pixels ONE launch (csrc/conv_small.hip, v3 / v4), otherwise
    (v2's softmax, many pixels) convolution + yolo_head_act_fwd behind the same entry. Against those two calls: t to fp32
    summation order, y to the activation's conditioning (yolov3/models/__init__.py:34-64)."""
    from tf2_yolo_amd import ops
    ops.ensure_conv_workspace()
    g = torch.Generator().manual_seed(31)
    cout = A * (5 + C)
    x = torch.randn(n, hw, hw, cin, generator=g).float().cuda()
    wk = (torch.randn(cout, cin, generator=g) / cin ** 0.5).float().cuda()
    b = (torch.randn(cout, generator=g) * 0.5).float().cuda()
    anchors = (torch.rand(A, 2, generator=g) + 0.1).float().cuda()
    d = ops.conv_desc((n, hw, hw, cin), cout, 1, 1, 1, "same")
    xp = ops.split_planes(x, n * hw * hw, cin)
    wp = ops.split_planes(wk, cout, cin)
    t_ref = ops.conv2d_fwd_planes(d, xp, wp, b)
    y_ref = ops.head_act_fwd(t_ref, A, C, version, anchors)
    t = torch.full_like(t_ref, float("nan"))
    y = torch.full_like(t_ref, float("nan"))
    ops.conv2d_fwd_head_unit(d, xp, wp, b, A, C, version, anchors, t, y)
    torch.cuda.synchronize()
    assert _relerr(t.double(), t_ref.double()) < 2e-6
    ref64 = x.double().reshape(-1, cin) @ wk.double().t() + b.double()
    assert _relerr(t.double().reshape(-1, cout), ref64) < 2e-6
    # y: exactly the head activation of the t this call wrote
    assert torch.equal(y, ops.head_act_fwd(t, A, C, version, anchors))
    assert torch.allclose(y, y_ref, rtol=2e-5, atol=1e-6)


@pytest.mark.parametrize("shape,act,with_y", [((2, 37, 45), "leaky", True), ((1, 64, 64), "mish", False),
                                              ((1, 416, 416), "leaky", False), ((3, 21, 130), "leaky", True)])
def test_stem_inference_unit(shape, act, with_y):
    """yolo_stem_fwd_infer_unit: Conv2D(32, 3, same) + folded BatchNorm + activation of the RGB image in one launch against
    the fp64 oracle; the planes hold the result to the format's 22 bits under the a-priori bound K max|image| + D, one word
    of max|result| per workgroup"""
    from tf2_yolo_amd import ops
    n, h, w = shape
    g = torch.Generator().manual_seed(31 + h)
    img = torch.rand(n, h, w, 3, generator=g, dtype=torch.float64)
    wk = torch.randn(3, 3, 3, 32, generator=g, dtype=torch.float64) * 0.3
    b = torch.randn(32, generator=g, dtype=torch.float64) * 0.1
    scale = (torch.rand(32, generator=g, dtype=torch.float64) + 0.5) * torch.where(torch.rand(32, generator=g) < 0.2, -1.0, 1.0)
    shift = torch.randn(32, generator=g, dtype=torch.float64) * 0.3
    z = L.conv2d(img, wk, b, stride=1, padding="same") * scale + shift
    ref = torch.where(z > 0, z, 0.1 * z) if act == "leaky" else z * torch.tanh(torch.nn.functional.softplus(z))
    d = ops.conv_desc((n, h, w, 3), 32, 3, 3, 1, "same")
    xd, wd, bd = img.float().cuda(), _krsc(wk).float().cuda(), b.float().cuda()
    sc, sh = scale.float().cuda(), shift.float().cuda()
    wt = torch.empty(28 * 32, device="cuda")
    ops.stem_filter_prep(wd, bd, wt)
    pred = torch.zeros(2, device="cuda")
    ops.conv_pred_bound(wd, 32, 27, sc, sh, bd, pred)
    words = torch.full((ops.INFER_BOUND_WORDS,), -1, device="cuda", dtype=torch.int32)
    n_in = ops.absmax_words(xd, words)
    assert 1 <= n_in <= 256 and float(words[:n_in].view(torch.float32).max()) == float(xd.abs().max())
    rows = n * h * w
    pl = torch.zeros(ops.planes_bytes(rows, 32), device="cuda", dtype=torch.uint8)
    out_words = torch.full((ops.INFER_BOUND_WORDS,), -1, device="cuda", dtype=torch.int32)
    y = torch.empty(n, h, w, 32, device="cuda") if with_y else None
    epi = ops.EPI_AFFINE_LEAKY if act == "leaky" else ops.EPI_AFFINE_MISH
    nw = ops.stem_fwd_infer_unit(d, xd, wt, epi, sc, sh, pred, words[:n_in], y, pl, out_words)
    torch.cuda.synchronize()
    vals, bound, s_, tail = _planes_values(pl.cpu(), rows, 32)
    refm = float(ref.abs().max())
    assert _relerr(vals.reshape(ref.shape), ref) < 1e-5
    assert bound >= refm and (tail == 0).all() and 2.0 ** 14 < s_ * bound <= 2.0 ** 15
    got_max = float(out_words[:nw].view(torch.float32).max())
    assert nw >= 1 and abs(got_max - refm) <= 1e-5 * refm and bool((out_words[nw:] == -1).all())
    if with_y:
        assert _relerr(y.double().cpu(), ref) < 1e-5
        yv = y.double().cpu().reshape(rows, 32)
        assert ((vals - yv).abs() <= torch.maximum(2.0 ** -22 * yv.abs(), torch.tensor(2.0 ** -25 / s_, dtype=torch.float64))).all()
        assert got_max == float(y.abs().max())


@pytest.mark.parametrize("shape", [(2, 37, 45), (1, 16, 16), (3, 130, 127), (2, 21, 400), (1, 50, 50)])
@pytest.mark.parametrize("act", ["leaky", "mish"])
def test_stem_backward_fused(shape, act):
    """yolo_stem_bn_bwd_wgrad (csrc/stem.hip): the BatchNorm / activation backward apply and the filter gradient of the
    stem unit in one pass. Against the unfused device path (yolo_bn_act_bwd_apply, then yolo_conv2d_wgrad on the dy it
    wrote) and, for the filter gradient, against float64 autograd on that dy."""
    from tf2_yolo_amd import ops
    from tf2_yolo_amd._lib import ACT_LEAKY, ACT_MISH
    a = ACT_LEAKY if act == "leaky" else ACT_MISH
    n, h, w = shape
    g = torch.Generator().manual_seed(81)
    img = torch.rand(n, h, w, 3, generator=g).cuda()
    y = torch.randn(n, h, w, 32, generator=g).cuda()
    dout = torch.randn(n, h, w, 32, generator=g).cuda()
    gamma = (1 + 0.2 * torch.randn(32, generator=g)).cuda()
    mean = (0.1 * torch.randn(32, generator=g)).cuda()
    inv = (0.5 + torch.rand(32, generator=g)).cuda()
    scale = gamma * inv
    shift = (0.1 * torch.randn(32, generator=g)).cuda() - mean * scale
    d = ops.conv_desc((n, h, w, 3), 32, 3, 3, 1, "same")

    def bufs():
        return (torch.zeros((ops.BN_RED_SLOTS + 1) * 64, device="cuda", dtype=torch.float64), torch.zeros(32, device="cuda"),
                torch.zeros(32, device="cuda"), torch.zeros(32 * 27, device="cuda"))

    red0, dg0, db0, dw0 = bufs()
    dy = ops.bn_act_bwd(y, dout, 32, gamma, scale, shift, mean, inv, a, red0, dg0, db0)
    ops.conv2d_wgrad(d, img, dy, dw0)
    red1, dg1, db1, dw1 = bufs()
    dw1 += 1.0    # accumulated, not overwritten
    ops.stem_bn_bwd_wgrad(d, img, y, dout, scale, shift, mean, inv, a, red1, dg1, db1, dw1)
    torch.cuda.synchronize()
    assert torch.equal(dg0, dg1) and torch.equal(db0, db1)
    assert _relerr((dw1 - 1.0).double(), dw0.double()) < 1e-5
    wk = torch.zeros(3, 3, 3, 32, dtype=torch.float64, requires_grad=True)
    out = L.conv2d(img.double().cpu(), wk, None, stride=1, padding="same")
    (out * dy.double().cpu()).sum().backward()
    assert _relerr((dw1 - 1.0).double().cpu().reshape(32, 27), _krsc(wk.grad).reshape(32, 27)) < TOL


def test_conv_planes_rejects_unsupported_shapes():
    from tf2_yolo_amd import ops
    from tf2_yolo_amd._lib import YoloHipError
    with pytest.raises(YoloHipError):
        ops.planes_bytes(10, 24)          # C % 16 != 0
    d = ops.conv_desc((1, 8, 8, 32), 16, 3, 3, 1, "same")   # Cout = 16: not covered by the planes kernels
    xp = ops.split_planes(torch.zeros(64, 32, device="cuda"), 64, 32)
    wp = ops.split_planes(torch.zeros(16, 288, device="cuda"), 16, 288)
    with pytest.raises(YoloHipError):
        ops.conv2d_fwd_planes(d, xp, wp)


# Cin % 16 == 0, Cout % 16 == 0, Cout >= 32, k*k*Cin >= 64
WGRAD_PLANES_CASES = [
    (2, 16, 16, 32, 64, 3, 1, "same", False),         # 64 x 128 tile
    (2, 17, 13, 32, 64, 3, 2, "darknet_s2", False),    # stride 2, odd sizes, pixel count not a multiple of 16
    (2, 16, 16, 48, 128, 3, 2, "darknet_s2", True),    # Cin = 48: column blocks straddle taps; bias
    (2, 7, 7, 96, 160, 3, 1, "same", False),          # Cout tail (160 = 128 + 32), column tail
    (3, 9, 11, 256, 128, 1, 1, "valid", False),
    (1, 2, 2, 1024, 512, 3, 1, "same", False),        # 4 pixels
    (2, 14, 14, 64, 64, 1, 1, "same", True),          # 64 x 64 tile
    (2, 13, 13, 64, 128, 1, 1, "same", False),        # 128 x 64 tile
    (4, 26, 26, 128, 256, 3, 1, "same", False),       # several pixel chunks (split-K) per tile
    (2, 24, 24, 64, 32, 1, 1, "same", False),         # Cout = 32: half of a 64-row tile (the 208x208 1x1 64 -> 32 layer)
    (4, 72, 72, 32, 64, 3, 2, "darknet_s2", False),   # stride 2 at 5184 output rows
    (2, 104, 104, 32, 64, 3, 2, "same", True),        # stride 2, Keras 'same' (pad after only), 5408 rows; bias
]


@pytest.mark.parametrize("case", WGRAD_PLANES_CASES)
def test_conv_wgrad_planes(case):
    from tf2_yolo_amd import ops
    n, h, w, cin, cout, k, s, pad, bias = case
    x, wk, b = _mk(case, seed=12)
    wk.requires_grad_(True)
    if b is not None:
        b.requires_grad_(True)
    ref = L.conv2d(x, wk, b, stride=s, padding=pad)
    g = torch.Generator().manual_seed(13)
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(dy)
    d = ops.conv_desc((n, h, w, cin), cout, k, k, s, pad)
    xp = ops.split_planes(x.float().cuda(), n * h * w, cin)
    dyd = dy.float().cuda()
    dyp = ops.split_planes(dyd, n * d.Ho * d.Wo, cout)
    dw = torch.zeros(cout, k, k, cin, device="cuda")
    db = torch.zeros(cout, device="cuda") if bias else None
    ops.conv2d_wgrad_planes(d, xp, dyp, dw, dy=dyd, dbias=db)
    torch.cuda.synchronize()
    assert _relerr(dw.double().cpu(), _krsc(wk.grad)) < TOL
    if bias:
        assert _relerr(db.double().cpu(), b.grad) < TOL
    ops.conv2d_wgrad_planes(d, xp, dyp, dw)       # accumulates: dw += ...
    torch.cuda.synchronize()
    assert _relerr(dw.double().cpu(), 2 * _krsc(wk.grad)) < TOL


@pytest.mark.parametrize("case", [WGRAD_PLANES_CASES[i] for i in (1, 2, 3, 8, 9, 11)] +
                         [(8, 52, 52, 128, 256, 3, 1, "same", True)])     # a benchmark layer at bs 8: 18 tiles x 42 splits
def test_conv_wgrad_planes_reproducible(case):
    """With the wgrad workspace registered (what Network does) the filter and bias gradients use no atomics: the splits of
    the pixel contraction are stored to slabs and added IN ORDER (wgrad_reduce_kernel, colsum_finish_kernel). Two runs are
    bit-identical, the result accumulates into dw like the atomic form, agrees with it to fp32 summation order and with
    the float64 oracle to 1e-4; YOLO_WGRAD_DETERMINISTIC is not consulted per call, so the atomic reference is taken first."""
    from tf2_yolo_amd import _lib, ops
    n, h, w, cin, cout, k, s, pad, bias = case
    x, wk, b = _mk(case, seed=12)
    wk.requires_grad_(True)
    if b is not None:
        b.requires_grad_(True)
    ref = L.conv2d(x, wk, b, stride=s, padding=pad)
    g = torch.Generator().manual_seed(13)
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(dy)
    d = ops.conv_desc((n, h, w, cin), cout, k, k, s, pad)
    xp = ops.split_planes(x.float().cuda(), n * h * w, cin)
    dyd = dy.float().cuda()
    dyp = ops.split_planes(dyd, n * d.Ho * d.Wo, cout)

    def run():
        dw = torch.full((cout, k, k, cin), 0.5, device="cuda")       # accumulates onto what is there
        db = torch.full((cout,), 0.25, device="cuda") if bias else None
        ops.conv2d_wgrad_planes(d, xp, dyp, dw, dy=dyd, dbias=db)
        torch.cuda.synchronize()
        return dw, db
    lib = _lib.load()
    _lib.check(lib.yolo_set_wgrad_workspace(None, 0), "yolo_set_wgrad_workspace")     # atomics
    ops._WGRAD_WS = None
    dw_a, db_a = run()
    ops.ensure_wgrad_workspace()
    dw1, db1 = run()
    dw2, db2 = run()
    assert torch.equal(dw1, dw2)
    assert _relerr((dw1 - 0.5).double().cpu(), _krsc(wk.grad)) < TOL
    assert _relerr(dw1.double(), dw_a.double()) < 1e-5
    if bias:
        assert torch.equal(db1, db2)
        assert _relerr((db1 - 0.25).double().cpu(), b.grad) < TOL and _relerr(db1.double(), db_a.double()) < 1e-5


# ---- 3x3 stride-1 filter gradient with the input window in LDS (csrc/conv_wgrad_win.hip; yolo_set_option key 6) ----
# Cout % 128 == 0, Cin % 32 == 0, 3x3 stride 1 'same', 32 / W + 2 <= H, 2 W + 81 <= 512
WGRAD_WIN_CASES = [
    (4, 26, 26, 128, 256, 3, 1, "same", False),       # 8 tiles, several splits
    (2, 13, 13, 64, 128, 3, 1, "same", True),         # stages that cross image rows and images; bias
    (3, 8, 7, 32, 128, 3, 1, "same", False),          # 56-pixel images: every stage crosses rows, most cross an image
    (1, 19, 38, 64, 128, 3, 1, "same", False),        # H != W
    (5, 13, 13, 32, 256, 3, 1, "same", False),        # 845 pixels: not a multiple of 16 (zero rows of the last planes block)
    (2, 76, 76, 32, 128, 3, 1, "same", False),        # the longest rows the 256-slot ring takes (2 W + 81 <= 256 up to 87)
    (1, 104, 104, 32, 128, 3, 1, "same", False),      # 512-slot ring
    (8, 52, 52, 128, 256, 3, 1, "same", False),       # a benchmark layer at bs 8
]


@pytest.mark.parametrize("variant", [1, 3, 4])   # 1: v_mfma_f32_32x32x16_f16, 3: on v_mfma_f32_16x16x32_f16, 4: two pixel halves per workgroup (half the slabs)
@pytest.mark.parametrize("case", WGRAD_WIN_CASES)
def test_conv_wgrad_window_kernel(case, variant):
    """wgrad_win_kernel: all nine taps of 32 input channels from ONE ring of x pixels in LDS, borders by pointing invalid
    (pixel, tap) pairs at a zero slot. Against the float64 oracle (1e-4), against the per-tap kernel (fp32 summation order),
    accumulation into dw, and -- with the workspace registered -- bit-identical from run to run."""
    from tf2_yolo_amd import _lib, ops
    n, h, w, cin, cout, k, s, pad, bias = case
    x, wk, b = _mk(case, seed=21)
    wk.requires_grad_(True)
    ref = L.conv2d(x, wk, None, stride=s, padding=pad)
    g = torch.Generator().manual_seed(22)
    dy = torch.randn(ref.shape, generator=g, dtype=torch.float64)
    ref.backward(dy)
    want = _krsc(wk.grad)
    d = ops.conv_desc((n, h, w, cin), cout, k, k, s, pad)
    xp = ops.split_planes(x.float().cuda(), n * h * w, cin)
    dyd = dy.float().cuda()
    dyp = ops.split_planes(dyd, n * d.Ho * d.Wo, cout)
    lib = _lib.load()

    def run(win, fill=0.0):
        ops.set_option(ops.OPT_WGRAD_WIN, win)
        dw = torch.full((cout, k, k, cin), fill, device="cuda")
        db = torch.zeros(cout, device="cuda") if bias else None
        ops.conv2d_wgrad_planes(d, xp, dyp, dw, dy=dyd, dbias=db)
        torch.cuda.synchronize()
        return dw, db
    try:
        _lib.check(lib.yolo_set_wgrad_workspace(None, 0), "yolo_set_wgrad_workspace")     # atomics
        ops._WGRAD_WS = None
        dw_old, _ = run(0)
        dw_at, db_at = run(variant)
        assert _relerr(dw_at.double().cpu(), want) < TOL
        assert _relerr(dw_at.double(), dw_old.double()) < 1e-5
        if bias:
            assert _relerr(db_at.double().cpu(), dy.sum((0, 1, 2))) < TOL
        ops.ensure_wgrad_workspace()                                                      # slabs + ordered reduce
        dw1, _ = run(variant, 0.5)
        dw2, _ = run(variant, 0.5)
        assert torch.equal(dw1, dw2)
        assert _relerr((dw1 - 0.5).double().cpu(), want) < TOL
        # border handling, element by element: the corner taps of the first / last filter rows
        err = ((dw1 - 0.5).double().cpu() - want).abs().amax(dim=(0, 3)) / want.abs().amax()
        assert float(err.max()) < TOL, err
    finally:
        ops.reset_options()


def test_batched_filter_split_equals_per_tensor_split():
    """yolo_split_planes_batch / yolo_filter_transpose_batch (one launch for every filter of a network) against the
    per-tensor entry points, byte for byte: ragged row counts, a job smaller than one workgroup, and a transposed
    filter that takes its bound from the planes of the untransposed one (same values -> same scale -> same bytes)."""
    from tf2_yolo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(5)
    shapes = [(255, 1, 512), (64, 9, 32), (16, 1, 16), (1024, 9, 512), (33, 9, 48)]   # (cout, taps, cin)
    ws = [torch.randn(co, t, ci, device="cuda", generator=g) * (0.01 + i) for i, (co, t, ci) in enumerate(shapes)]
    wTs = [torch.empty(ci * t * co, device="cuda") for (co, t, ci) in shapes]
    jt = ops.BatchJobs("transpose", "cuda")
    for w, wT, (co, t, ci) in zip(ws, wTs, shapes):
        jt.add_transpose(w, wT, co, t, ci)
    jt.run()
    for w, wT, (co, t, ci) in zip(ws, wTs, shapes):
        assert torch.equal(wT.view(ci, t, co), w.permute(2, 1, 0).contiguous())
    fwd = [torch.zeros(ops.planes_bytes(co, t * ci), dtype=torch.uint8, device="cuda") for (co, t, ci) in shapes]
    js = ops.BatchJobs("split", "cuda")
    for w, p, (co, t, ci) in zip(ws, fwd, shapes):
        js.add_split(w, p, co, t * ci)
    js.run()
    def same(a, b):   # body + {bound, scale, 1/scale}; the rest of the 256-byte header is unused
        n = a.numel() - 256 + 12
        return torch.equal(a[:n], b[:n])
    for w, p, (co, t, ci) in zip(ws, fwd, shapes):
        assert same(p, ops.split_planes(w, co, t * ci)), (co, t, ci)
    # transposed filters [cin][taps*cout] need taps*cout % 16 == 0: skip the 255- and 33-filter layers
    sel = [i for i, (co, t, ci) in enumerate(shapes) if (t * co) % 16 == 0]
    bwd = {i: torch.zeros(ops.planes_bytes(shapes[i][2], shapes[i][1] * shapes[i][0]), dtype=torch.uint8, device="cuda")
           for i in sel}
    jb = ops.BatchJobs("split", "cuda")
    for n, i in enumerate(sel):
        co, t, ci = shapes[i]
        jb.add_split(wTs[i], bwd[i], ci, t * co, bound_from=fwd[i] if n % 2 == 0 else None)
    jb.run()
    for i in sel:
        co, t, ci = shapes[i]
        assert same(bwd[i], ops.split_planes(wTs[i], ci, t * co)), shapes[i]


def test_split_planes_padded_equals_split_of_the_zero_padded_tensor():
    from tf2_yolo_amd import ops
    g = torch.Generator(device="cuda").manual_seed(9)
    for rows, c in [(5408, 255), (37, 125), (16, 1), (1000, 250)]:
        x = torch.randn(rows, c, device="cuda", generator=g) * 3.0
        cp = (c + 15) // 16 * 16
        xz = torch.zeros(rows, cp, device="cuda")
        xz[:, :c] = x
        a, b = ops.split_planes_padded(x, rows, c), ops.split_planes(xz, rows, cp)
        n = a.numel() - 256 + 12
        assert a.numel() == b.numel() and torch.equal(a[:n], b[:n]), (rows, c)
    with pytest.raises(Exception):
        ops.split_planes_padded(torch.zeros(4, 256, device="cuda"), 4, 250)


# ---- BatchNormalization-backward reduction fused into the data gradient that completes dL/d(a) (round 5:
# yolo_conv2d_dgrad_planes_bnred + yolo_bn_act_bwd_sum_partials; planes_epilogue.hpp) ----
BNRED_CASES = [
    # (N, H, W, Cin, Cout, k, stride, padding, bias), forced options {key: value}
    ((2, 20, 17, 128, 256, 3, 1, "same", False), {}),                       # window kernel, 128 columns
    ((3, 13, 13, 64, 128, 3, 1, "same", False), {}),                        # per-tap kernel (Cin = 64 columns), tiles cross images
    ((2, 26, 26, 256, 128, 1, 1, "same", False), {}),                       # 1x1: the residual blocks' accumulate form
    ((2, 16, 16, 32, 64, 1, 1, "same", False), {}),                         # 1x1 into 32 channels: the 128 x 32 tile
    ((2, 17, 13, 32, 64, 3, 2, "darknet_s2", False), {}),                   # stride 2: four parity classes in one launch, odd sizes
    ((2, 16, 16, 64, 128, 3, 2, "darknet_s2", False), {}),
    ((2, 14, 14, 64, 64, 3, 2, "same", False), {}),                         # v1.5: 3x3 s2 'same' (asymmetric pad)
    ((2, 40, 72, 128, 64, 3, 1, "same", False), {5: 2}),                    # 2-D patch window (128 x 128 tiles on 8 x 16 patches)
    ((1, 152, 152, 64, 64, 3, 1, "same", False), {5: 2}),                   # patch, 256 x 64 tiles, ragged patches
    ((1, 104, 104, 64, 128, 3, 1, "same", False), {}),                      # > 256 slots: the sum kernel's chunks + last arriver
    ((32, 13, 13, 512, 1024, 3, 1, "same", False), {}),                     # a benchmark layer (split-K is off for the fused form)
]


@pytest.mark.parametrize("act", [1, 2])
@pytest.mark.parametrize("case,opts", BNRED_CASES)
def test_dgrad_fused_bn_backward_reduction(case, opts, act):
    """The fused form against the two separate passes it replaces, on the same inputs: dx BIT-identical to
    yolo_conv2d_dgrad_planes (plain and accumulate form), the folded sums of dz and dz * xhat within 2e-6 of the standalone
    reduction's (fp32 tile sums folded in fp64 against per-element fp64 accumulation; relative to sum |dz|), max|dz| and the
    apply step's output (dx of the BatchNorm, its planes, dgamma, dbeta) within fp32 rounding of the standalone path's, and the
    whole thing bit-identical run to run (no atomics in the sums)."""
    from tf2_yolo_amd import ops
    from tf2_yolo_amd._lib import ACT_LEAKY
    ops.ensure_conv_workspace()
    n, h, w, cin, cout, k, s, pad, bias = case
    g = torch.Generator(device="cuda").manual_seed(100 + cin + cout + k + s)
    d = ops.conv_desc((n, h, w, cin), cout, k, k, s, pad)
    P, C = n * h * w, cin
    wk = torch.randn(cout, k, k, cin, device="cuda", generator=g) / (k * k * cin) ** 0.5
    wT = ops.filter_transpose(wk, cout, k * k, cin)
    dy = torch.randn(n, d.Ho, d.Wo, cout, device="cuda", generator=g)
    dyp = ops.split_planes(dy, n * d.Ho * d.Wo, cout)
    wTp = ops.split_planes(wT, cin, k * k * cout)
    y = torch.randn(P, C, device="cuda", generator=g) * 1.5 + 0.2          # pre-BN tensor of the producer
    gamma = torch.rand(C, device="cuda", generator=g) + 0.5
    smean, var = y.mean(0), y.var(0, unbiased=False)
    sinv = (1.0 / torch.sqrt(var + 1e-3)).contiguous()
    scale = (gamma * sinv).contiguous()
    shift = (torch.randn(C, device="cuda", generator=g) * 0.1 - smean * scale).contiguous()
    old = torch.randn(n, h, w, cin, device="cuda", generator=g)            # what the accumulate form adds into
    # (split-K off: the fused form never splits -- its reduction lives in the tile epilogue -- and the bit-identity of dx is
    # between the same kernels)
    ops.set_option(ops.OPT_CONV_SK, 0)
    for key, val in opts.items():
        ops.set_option(key, val)
    try:
        for accumulate in (False, True):
            def standalone():
                dx = old.clone() if accumulate else None
                dx = ops.conv2d_dgrad_planes(d, dyp, wTp, dx=dx, accumulate=accumulate)
                red = torch.zeros(513 * 2 * C, device="cuda", dtype=torch.float64)
                aux = torch.zeros(68, device="cuda", dtype=torch.int32)
                dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
                pl = torch.zeros(ops.planes_bytes(P, C), device="cuda", dtype=torch.uint8) if C % 16 == 0 else None
                dyy = ops.bn_act_bwd(y, dx, C, gamma, scale, shift, smean, sinv, act, red, dg, db, planes=pl,
                                     bound_aux=aux if pl is not None else None)
                torch.cuda.synchronize()
                return dx, red[512 * 2 * C:].clone(), aux, dg, db, pl, dyy

            def fused():
                cap = ops.bnred_slots_cap(d)
                part = torch.full((cap * 2 * C,), float("nan"), device="cuda")   # every slot used must have been written
                aux = torch.zeros(68, device="cuda", dtype=torch.int32)
                b = ops.BnReduce(y, scale, shift, smean, sinv, act, part, cap, aux)
                dx = old.clone() if accumulate else None
                dx = ops.conv2d_dgrad_planes(d, dyp, wTp, dx=dx, accumulate=accumulate, bnred=b)
                assert 0 < b.nslots <= cap
                red = torch.zeros(513 * 2 * C, device="cuda", dtype=torch.float64)
                dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
                pl = torch.zeros(ops.planes_bytes(P, C), device="cuda", dtype=torch.uint8) if C % 16 == 0 else None
                dyy = ops.bn_act_bwd(y, dx, C, gamma, scale, shift, smean, sinv, act, red, dg, db, planes=pl, bound_aux=aux,
                                     fused=b)
                torch.cuda.synchronize()
                assert int(aux[3]) == 0                                          # the ticket word is back at zero
                return dx, red[512 * 2 * C:].clone(), aux, dg, db, pl, dyy

            s_dx, s_red, s_aux, s_dg, s_db, s_pl, s_dyy = standalone()
            f_dx, f_red, f_aux, f_dg, f_db, f_pl, f_dyy = fused()
            assert torch.equal(s_dx, f_dx)
            # reference for the sums' scale: sum |dz| per channel, from the standalone tensors in float64
            z = scale.double() * y.double() + shift.double()
            if act == ACT_LEAKY:
                dact = torch.where(z > 0, 1.0, 0.1)
            else:
                sp = torch.nn.functional.softplus(z)
                t = torch.tanh(sp)
                dact = t + z * (1 - t * t) * torch.sigmoid(z)
            dz = s_dx.reshape(P, C).double() * dact
            xh = (y.double() - smean.double()) * sinv.double()
            ref = torch.stack([dz.sum(0), (dz * xh).sum(0)]).reshape(-1)
            mag = torch.stack([dz.abs().sum(0), (dz * xh).abs().sum(0)]).reshape(-1)
            assert ((f_red - ref).abs() / mag).max().item() < 2e-6
            assert ((s_red - ref).abs() / mag).max().item() < 2e-6
            assert f_aux.view(torch.float32)[0].item() == pytest.approx(dz.abs().max().item(), rel=1e-5)
            assert s_aux.view(torch.float32)[0].item() == pytest.approx(f_aux.view(torch.float32)[0].item(), rel=1e-6) or s_pl is None
            scl = max(s_dyy.abs().max().item(), 1e-30)
            assert (s_dyy - f_dyy).abs().max().item() / scl < 1e-5
            assert (s_dg - f_dg).abs().max().item() / max(s_dg.abs().max().item(), 1e-30) < 1e-5
            assert (s_db - f_db).abs().max().item() / max(s_db.abs().max().item(), 1e-30) < 1e-5
            again = fused()
            for a_, b_ in zip((f_dx, f_red, f_dg, f_db, f_pl, f_dyy), (again[0], again[1], again[3], again[4], again[5], again[6])):
                assert (a_ is None and b_ is None) or torch.equal(a_, b_)
    finally:
        ops.reset_options()
