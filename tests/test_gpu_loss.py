"""GPU parity of the fused loss (+gradient) and metrics kernels against the torch-CPU float64
restatement of the reference closures (oracle/losses.py, oracle/metrics.py), plus the closed-form
known-answer tests of SURVEY.md section 4.3. Tolerance 1e-4 (north_star)."""
import numpy as np
import pytest
import torch

from oracle import losses as OL
from oracle import metrics as OM

pytestmark = pytest.mark.gpu
TOL = 1e-4

ANCH9 = [[0.89663461, 0.78365384], [0.375, 0.47596153], [0.27884615, 0.21634615], [0.14182692, 0.28605769],
         [0.14903846, 0.10817307], [0.07211538, 0.14663461], [0.07932692, 0.05528846], [0.03846153, 0.07211538],
         [0.02403846, 0.03125]]


def make_case(N, g, A, C, anchors, seed, v1=False, obj_frac=0.15):
    rng = np.random.default_rng(seed)
    yt = np.zeros((N, g, g, 5 + C), dtype=np.float32)
    mask = rng.random((N, g, g)) < obj_frac
    n = int(mask.sum())
    yt[mask, 0:2] = rng.random((n, 2))
    yt[mask, 2:4] = rng.random((n, 2)) * 0.5 + 0.03
    yt[mask, 4] = 1
    cls = rng.integers(0, C, n)
    tmp = np.zeros((n, C), dtype=np.float32)
    tmp[np.arange(n), cls] = 1
    yt[mask, 5:] = tmp
    if v1:
        yp = rng.random((N, g, g, 5 * A + C)).astype(np.float32) * 0.98 + 0.01
        e = np.exp(rng.standard_normal((N, g, g, C)))
        yp[..., 5 * A:] = e / e.sum(-1, keepdims=True)
    else:
        yp = np.zeros((N, g, g, A, 5 + C), dtype=np.float32)
        yp[..., 0:2] = rng.random((N, g, g, A, 2))
        anc = np.array(anchors, dtype=np.float32).reshape(1, 1, 1, A, 2)
        yp[..., 2:4] = np.exp(rng.standard_normal((N, g, g, A, 2)) * 0.5) * anc
        yp[..., 4] = rng.random((N, g, g, A))
        yp[..., 5:] = rng.random((N, g, g, A, C)) * 0.98 + 0.01
        # make some predictions overlap their truth strongly so the ignore / truth thresholds fire
        sel = mask & (rng.random((N, g, g)) < 0.5)
        yp[sel, 0, 0:4] = yt[sel, 0:4] * (1 + 0.05 * rng.standard_normal((int(sel.sum()), 4))).astype(np.float32)
        yp = yp.reshape(N, g, g, A * (5 + C))
    return yt, yp


def run_gpu(cfg, yt, yp, grad_scale=1.0):
    from tf2_yolo_amd import ops
    out, dp = ops.loss_fwd_bwd(cfg, torch.tensor(yt).cuda(), torch.tensor(yp).cuda(), grad_scale=grad_scale)
    torch.cuda.synchronize()
    return out.cpu().numpy(), dp.cpu().double()


def check(loss_fn, cfg, yt, yp):
    ypt = torch.tensor(yp, dtype=torch.float64, requires_grad=True)
    ref = loss_fn(torch.tensor(yt, dtype=torch.float64), ypt)
    ref.backward()
    out, dp = run_gpu(cfg, yt, yp)
    assert abs(out[0] - ref.item()) <= TOL * max(abs(ref.item()), 1.0), (out[0], ref.item())
    gscale = ypt.grad.abs().max().item()
    assert (dp - ypt.grad).abs().max().item() <= TOL * gscale, ((dp - ypt.grad).abs().max().item(), gscale)


# (gamma = 2, the reference's default, takes the x * x form of the focal terms; 1.5 keeps libm's powf: both are checked)
@pytest.mark.parametrize("focal,use_scale,level,gamma", [(False, True, 0, 2), (True, True, 1, 2), (False, False, 2, 2),
                                                          (True, False, 0, 2), (True, True, 0, 1.5)])
def test_v3_loss(focal, use_scale, level, gamma):
    from tf2_yolo_amd import ops
    g = 13 * 2 ** level if level < 2 else 20
    N, A, C = 3, 3, 6
    anchors = ANCH9[3 * level:3 * level + 3]
    yt, yp = make_case(N, g, A, C, anchors, seed=10 + level)
    kw = dict(binary_weight=0.7, loss_weight=[1.5, 1.2, 5, 0.8], ignore_thresh=0.6, use_focal_loss=focal,
              focal_loss_gamma=gamma, use_scale=use_scale)
    fn = OL.wrap_yolo_loss_v3((g, g), A, C, anchors=anchors, **kw)
    cfg = ops.make_loss_cfg(3, N, g, g, A, C, anchors, binary_weight=0.7, loss_weight=[1.5, 1.2, 5, 0.8],
                            ignore_thresh=0.6, use_focal_loss=focal, focal_gamma=gamma, use_scale=use_scale)
    check(fn, cfg, yt, yp)


def test_v3_loss_c80_and_no_anchors():
    from tf2_yolo_amd import ops
    N, g, A, C = 2, 13, 3, 80
    yt, yp = make_case(N, g, A, C, ANCH9[:3], seed=3)
    check(OL.wrap_yolo_loss_v3((g, g), A, C, anchors=ANCH9[:3], loss_weight=[1, 1, 5, 1]),
          ops.make_loss_cfg(3, N, g, g, A, C, ANCH9[:3], loss_weight=[1, 1, 5, 1]), yt, yp)
    check(OL.wrap_yolo_loss_v3((g, g), A, C, anchors=None), ops.make_loss_cfg(3, N, g, g, A, C, None), yt, yp)


def test_v3_known_answers():
    """y_true = 0, xy=.5, wh=anchor, conf=c0 => loss = w2*bw*gh*gw*B*c0^2 = 633.75;
    batch-size invariance; argmax tie -> lowest anchor index."""
    from tf2_yolo_amd import ops
    g, A, C = 13, 3, 80
    anc = ANCH9[:3]
    for N in (1, 4):
        yt = np.zeros((N, g, g, 5 + C), dtype=np.float32)
        yp = np.zeros((N, g, g, A, 5 + C), dtype=np.float32)
        yp[..., 0:2] = .5
        yp[..., 4] = .5
        yp[..., 5:] = .3
        for b in range(A):
            yp[..., b, 2], yp[..., b, 3] = anc[b]
        cfg = ops.make_loss_cfg(3, N, g, g, A, C, anc, binary_weight=1, loss_weight=[1, 1, 5, 1])
        out, _ = run_gpu(cfg, yt, yp.reshape(N, g, g, -1))
        assert abs(out[0] - 633.75) < 1e-3
    # tie: two identical anchors/predictions -> anchor 0 is responsible
    yt = np.zeros((1, 1, 1, 5 + 2), dtype=np.float32)
    yt[0, 0, 0] = [.5, .5, .3, .3, 1, 1, 0]
    yp = np.zeros((1, 1, 1, 2, 7), dtype=np.float32)
    yp[..., :] = [.5, .5, .3, .3, .4, .6, .4]
    cfg = ops.make_loss_cfg(3, 1, 1, 1, 2, 2, [[.3, .3], [.3, .3]], loss_weight=[1, 1, 1, 1])
    out, dp = run_gpu(cfg, yt, yp.reshape(1, 1, 1, -1))
    dp = dp.reshape(2, 7)
    assert dp[0, 4] < 0 and dp[1, 4] == 0.0       # anchor 0 pulled up; anchor 1 ignored (iou 1 >= .6)


def test_v2_loss():
    from tf2_yolo_amd import ops
    N, g, A, C = 2, 13, 5, 20
    anchors = [(0.04405615, 0.05210654), (0.14418923, 0.15865615), (0.25680231, 0.42110308),
               (0.60637077, 0.27136769), (0.75157846, 0.70525231)]
    yt, yp = make_case(N, g, A, C, anchors, seed=21)
    # softmax-like class predictions
    ypr = yp.reshape(N, g, g, A, 5 + C)
    ypr[..., 5:] /= ypr[..., 5:].sum(-1, keepdims=True)
    fn = OL.wrap_yolo_loss_v2((g, g), A, C, anchors, binary_weight=0.5, loss_weight=[1, 1, 5, 1], ignore_thresh=.6)
    cfg = ops.make_loss_cfg(2, N, g, g, A, C, anchors, binary_weight=0.5, loss_weight=[1, 1, 5, 1], ignore_thresh=.6)
    check(fn, cfg, yt, yp)


@pytest.mark.parametrize("truth,smooth,gamma", [(1.0, 0.0, 2), (0.7, 0.0, 2), (1.0, 0.1, 2), (0.7, 0.05, 1.5)])
def test_v4_loss(truth, smooth, gamma):
    from tf2_yolo_amd import ops
    N, g, A, C = 2, 19, 3, 5
    anchors = [[0.75493421, 0.65953947], [0.31578947, 0.39967105], [0.23355263, 0.18092105]]
    yt, yp = make_case(N, g, A, C, anchors, seed=40 + int(truth * 10))
    fn = OL.wrap_yolo_loss_v4((g, g), A, C, anchors, binary_weight=0.9, loss_weight=[1, 5, 1], wh_reg_weight=0.01,
                              ignore_thresh=.6, truth_thresh=truth, label_smooth=smooth, focal_loss_gamma=gamma)
    cfg = ops.make_loss_cfg(4, N, g, g, A, C, anchors, binary_weight=0.9, loss_weight=[1, 5, 1], wh_reg_weight=0.01,
                            ignore_thresh=.6, truth_thresh=truth, label_smooth=smooth, focal_gamma=gamma)
    check(fn, cfg, yt, yp)


def test_v4_known_answer():
    """y_true=0 => loss = w1*gh*gw*B*(-c0^g*ln(1-c0)) (+0 reg with wh=anchor)."""
    from tf2_yolo_amd import ops
    g, A, C, N = 19, 3, 4, 2
    anc = [[0.75, 0.66], [0.31, 0.4], [0.23, 0.18]]
    yt = np.zeros((N, g, g, 5 + C), dtype=np.float32)
    yp = np.zeros((N, g, g, A, 5 + C), dtype=np.float32)
    yp[..., 0:2] = .5
    yp[..., 4] = .5
    yp[..., 5:] = .3
    for b in range(A):
        yp[..., b, 2], yp[..., b, 3] = anc[b]
    cfg = ops.make_loss_cfg(4, N, g, g, A, C, anc, loss_weight=[1, 5, 1], focal_gamma=2)
    out, _ = run_gpu(cfg, yt, yp.reshape(N, g, g, -1))
    expect = 5 * g * g * A * (-(0.5 ** 2) * np.log(0.5))
    assert abs(out[0] - expect) < 1e-4 * expect


@pytest.mark.parametrize("B,C,g", [(2, 1, 4), (2, 20, 7), (3, 4, 7)])
def test_v1_loss(B, C, g):
    from tf2_yolo_amd import ops
    N = 3
    yt, yp = make_case(N, g, B, C, None, seed=60 + C, v1=True, obj_frac=0.4)
    fn = OL.wrap_yolo_loss_v1((g, g), B, C, binary_weight=0.3, loss_weight=[5, 5, 1, 1])
    cfg = ops.make_loss_cfg(1, N, g, g, B, C, None, binary_weight=0.3, loss_weight=[5, 5, 1, 1])
    check(fn, cfg, yt, yp)


def test_loss_rejects_bad_shapes():
    from tf2_yolo_amd import ops
    from tf2_yolo_amd._lib import YoloHipError
    cfg = ops.make_loss_cfg(3, 1, 13, 13, 3, 2, ANCH9[:3])
    with pytest.raises(YoloHipError):
        ops.loss_fwd_bwd(cfg, torch.zeros(1, 13, 13, 7).cuda(), torch.zeros(1, 13, 13, 20).cuda())


@pytest.mark.parametrize("version", [3, 1])
def test_metrics(version):
    from tf2_yolo_amd import ops
    N, g, A, C = 3, 13, 3, 6
    if version == 1:
        A, C, g = 2, 5, 7
        yt, yp = make_case(N, g, A, C, None, seed=5, v1=True, obj_frac=0.4)
        cfg = ops.make_loss_cfg(1, N, g, g, A, C, None)
    else:
        yt, yp = make_case(N, g, A, C, ANCH9[:3], seed=5)
        cfg = ops.make_loss_cfg(3, N, g, g, A, C, ANCH9[:3])
    out = ops.metrics(cfg, torch.tensor(yt).cuda(), torch.tensor(yp).cuda(), recall_thresh=0.5).cpu().numpy()
    t, p = torch.tensor(yt, dtype=torch.float64), torch.tensor(yp, dtype=torch.float64)
    cells = N * g * g
    if version == 1:
        ref = [OM.obj_acc_v1(t, p, (g, g), A, C).mean().item(), OM.mean_iou_v1(t, p, (g, g), A, C).item(),
               OM.class_acc_v1(t, p, (g, g), C).item(), OM.recall_v1(t, p, (g, g), A, C, 0.5).item()]
        got = [out[0] / cells, out[1] / (out[2] + 1e-7), out[3] / (out[2] + 1e-7), out[4] / (out[2] + 1e-7)]
    else:
        ref = [OM.obj_acc(t, p, (g, g), A, C).mean().item(), OM.mean_iou(t, p, (g, g), A, C).item(),
               OM.class_acc(t, p, (g, g), A, C).item(), OM.recall(t, p, (g, g), A, C, 0.5).item()]
        got = [out[0] / cells, out[1] / (out[2] + 1e-7), out[3] / (out[2] * A + 1e-7), out[4] / (out[2] + 1e-7)]
    assert out[5] == cells
    for a, b in zip(got, ref):
        assert abs(a - b) < 1e-5, (got, ref)


@pytest.mark.parametrize("version,A,C,g,N", [(3, 3, 80, 13, 4), (3, 3, 80, 52, 2), (3, 3, 3, 8, 3), (4, 3, 80, 19, 2), (2, 5, 20, 13, 3),
                                             (3, 3, 1, 5, 1), (3, 3, 91, 6, 2)])
def test_loss_cell_ahead_loader_equals_chunk_ahead_loader(version, A, C, g, N):
    """Round 6: the loss kernel requests a whole cell (its <= 256 prediction channels, class targets, true box, the anchors'
    predicted boxes) one cell ahead; yolo_set_option(8, 8) keeps the round-4 loader (one 64-channel chunk ahead). Same
    arithmetic on the same values in the same per-lane order: the gradient (v2 / v3) and the exported decisions must be
    BIT-identical, the loss parts equal up to the order of the per-workgroup fp64 atomics. (C = 91: 288 channels per cell, more than the
    cell-ahead loader holds -- the launcher keeps the old loader; one cell per wave and fewer cells than waves are covered by
    the small grids.)"""
    from tf2_yolo_amd import ops
    anchors = ANCH9[:A] if A <= 9 else None
    if A == 5:
        anchors = [[0.75, 0.70], [0.60, 0.27], [0.25, 0.42], [0.14, 0.15], [0.04, 0.05]]
    yt, yp = make_case(N, g, A, C, anchors, seed=version * 100 + C)
    kw = dict(loss_weight=(1, 1, 5, 1)) if version != 4 else dict(loss_weight=(1, 5, 1), truth_thresh=0.7)
    cfg = ops.make_loss_cfg(version, N, g, g, A, C, anchors=anchors, **kw)
    ytd, ypd = torch.tensor(yt).cuda(), torch.tensor(yp).cuda()
    res = {}
    try:
        for opt in (8, 0):
            ops.set_option(8, opt)
            dec = torch.full((N * g * g, 2), -7, dtype=torch.int32, device="cuda")
            out, dp = ops.loss_fwd_bwd(cfg, ytd, ypd, decisions=dec)
            torch.cuda.synchronize()
            res[opt] = (out.cpu().numpy().copy(), dp.clone(), dec.clone())
    finally:
        ops.reset_options()
    if version == 4:   # (the CIoU chain is contracted into fused multiply-adds differently in the two instantiations: last bits)
        scale = res[8][1].abs().max().item()
        assert (res[0][1] - res[8][1]).abs().max().item() <= 1e-6 * scale
    else:
        assert torch.equal(res[0][1], res[8][1])
    assert torch.equal(res[0][2], res[8][2]) and int(res[0][2].min()) >= 0
    assert np.allclose(res[0][0], res[8][0], rtol=1e-7 if version == 4 else 1e-12, atol=0)
