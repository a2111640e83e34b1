"""The hand-derived known answers of tests/kat_cases.py through the C-ABI (fused loss / metrics kernels, conv
kernels incl. the planes path, space_to_depth, max-pool) -- the same cases the CPU oracle passes in
tests/test_kat_oracle_cpu.py. Tolerance: fp32 1e-4 on losses (float32 inputs, fp64 accumulation on the device),
exact for the integer-valued layer KATs."""
import numpy as np
import pytest
import torch

import kat_cases as K

pytestmark = pytest.mark.gpu


def _loss(cfg, yt, yp):
    from tf2_yolo_amd import ops
    out, _ = ops.loss_fwd_bwd(cfg, torch.tensor(yt).cuda(), torch.tensor(yp).cuda())
    torch.cuda.synchronize()
    return float(out[0].item())


@pytest.mark.parametrize("case", [K.v2_empty, K.v2_one_object])
def test_v2_loss_kat(case):
    from tf2_yolo_amd import ops
    c = case()
    cfg = ops.make_loss_cfg(2, c["N"], c["g"], c["g"], c["A"], c["C"], c["anchors"], **c["kw"])
    got = _loss(cfg, c["yt"], c["yp"])
    assert abs(got - c["expect"]) < 1e-4 * max(1.0, abs(c["expect"])), (got, c["expect"])


def test_v1_loss_kat():
    from tf2_yolo_amd import ops
    c = K.v1_one_object()
    cfg = ops.make_loss_cfg(1, 1, 1, 1, c["B"], c["C"], None, **c["kw"])
    got = _loss(cfg, c["yt"], c["yp"])
    assert abs(got - c["expect"]) < 1e-4, (got, c["expect"])


@pytest.mark.parametrize("name", list(K.CIOU_GEOMETRIES))
def test_ciou_kat(name):
    from tf2_yolo_amd import ops
    c = K.ciou_case(name)
    kw = dict(c["kw"])
    kw["focal_gamma"] = kw.pop("focal_loss_gamma")
    cfg = ops.make_loss_cfg(4, 1, 1, 1, 1, 1, c["anchors"], **kw)
    got = _loss(cfg, c["yt"], c["yp"])
    assert abs(got - c["expect"]) < 1e-4 * max(1.0, abs(c["expect"])), (got, c["expect"], c["ciou"])


def test_metrics_kats():
    from tf2_yolo_amd import ops
    for c, version, A in ((K.metrics_v3(), 3, None), (K.metrics_v1(), 1, None)):
        A = c.get("A", c.get("B"))
        cfg = ops.make_loss_cfg(version, c["N"], c["gh"], c["gw"], A, c["C"], [(0.5, 0.5)] * A if version == 3 else None)
        out = ops.metrics(cfg, torch.tensor(c["yt"]).cuda(), torch.tensor(c["yp"]).cuda(), recall_thresh=0.5).cpu().numpy()
        e = c["expect"]
        cells = c["N"] * c["gh"] * c["gw"]
        assert out[5] == cells and out[0] == sum(e["obj_acc"])          # sum over cells of binary_accuracy
        assert abs(out[1] / (out[2] + 1e-7) - e["mean_iou"]) < 1e-6
        denom = out[2] * A if version == 3 else out[2]
        assert abs(out[3] / (denom + 1e-7) - e["class_acc"]) < 1e-6
        assert abs(out[4] / (out[2] + 1e-7) - e["recall"]) < 1e-6


@pytest.mark.parametrize("c", K.conv_tap_cases())
def test_conv_padding_kat(c):
    """one-hot filters: the output must be the shifted input of the Keras padding rule, bit for bit (values are small
    integers), on the fp32-operand kernels AND on the planes kernels"""
    from tf2_yolo_amd import ops
    k, H, W = c["k"], c["H"], c["W"]
    cin, cout = 32, 32
    x = torch.zeros(1, H, W, cin)
    x[0, :, :, 0] = torch.tensor(c["x"])
    w = torch.zeros(cout, k, k, cin)
    w[1, c["tap"][0], c["tap"][1], 0] = 1.0
    d = ops.conv_desc((1, H, W, cin), cout, k, k, c["stride"], c["padding"])
    assert (d.Ho, d.Wo) == c["y"].shape
    xd, wd = x.cuda(), w.cuda()
    y = ops.conv2d_fwd(d, xd, wd)
    yp = ops.conv2d_fwd_planes(d, ops.split_planes(xd, H * W, cin), ops.split_planes(wd, cout, k * k * cin))
    torch.cuda.synchronize()
    for got in (y, yp):
        assert np.array_equal(got[0, :, :, 1].cpu().numpy(), c["y"])
        assert float(got[0, :, :, 0].abs().max()) == 0.0


def test_space_to_depth_kat():
    from tf2_yolo_amd import ops
    x, y = K.space_to_depth_case()
    out = torch.empty(1, 2, 2, 8, device="cuda")
    ops.space_to_depth2_fwd(torch.tensor(x).cuda(), out, 8, 0)
    torch.cuda.synchronize()
    assert np.array_equal(out.cpu().numpy(), y)


@pytest.mark.parametrize("c", K.maxpool_cases())
def test_maxpool_pad_kat(c):
    from tf2_yolo_amd import ops
    H, W = c["H"], c["W"]
    Ho, Wo = c["y"].shape
    x = torch.tensor(c["x"]).reshape(1, H, W, 1).repeat(1, 1, 1, 4).contiguous().cuda()
    out = torch.empty(1, Ho, Wo, 4, device="cuda")
    ops.maxpool_fwd(x, c["k"], c["stride"], c["pad_t"], c["pad_l"], Ho, Wo, out, 4, 0, None)
    torch.cuda.synchronize()
    for ch in range(4):
        assert np.array_equal(out[0, :, :, ch].cpu().numpy(), c["y"])
