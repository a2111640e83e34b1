"""The hand-derived known answers of tests/kat_cases.py against the CPU oracle (float64). The same cases run
through the C-ABI in tests/test_gpu_kats.py."""
import numpy as np
import pytest
import torch

import kat_cases as K
from oracle import layers as L
from oracle import losses as OL
from oracle import metrics as OM

T = lambda a: torch.tensor(a, dtype=torch.float64)


@pytest.mark.parametrize("case", [K.v2_empty, K.v2_one_object])
def test_v2_loss_kat(case):
    c = case()
    f = OL.wrap_yolo_loss_v2((c["g"], c["g"]), c["A"], c["C"], c["anchors"], **c["kw"])
    got = f(T(c["yt"]), T(c["yp"])).item()
    assert abs(got - c["expect"]) < 1e-6 * max(1.0, abs(c["expect"])), (got, c["expect"])


def test_v2_empty_value():
    assert K.v2_empty()["expect"] == 528.125      # 5 * 0.5 * 13 * 13 * 5 * 0.25


def test_v1_loss_kat():
    c = K.v1_one_object()
    f = OL.wrap_yolo_loss_v1((1, 1), c["B"], c["C"], **c["kw"])
    got = f(T(c["yt"]), T(c["yp"])).item()
    assert abs(got - c["expect"]) < 1e-6, (got, c["expect"])


@pytest.mark.parametrize("name", list(K.CIOU_GEOMETRIES))
def test_ciou_kat(name):
    c = K.ciou_case(name)
    t, p, _ = K.CIOU_GEOMETRIES[name]
    iou, ciou = OL.cal_iou(T([t]).reshape(1, 1, 4), T([p]).reshape(1, 1, 4), (1, 1), return_ciou=True)
    assert abs(ciou.item() - c["ciou"]) < 1e-9
    f = OL.wrap_yolo_loss_v4((1, 1), 1, 1, c["anchors"], **c["kw"])
    got = f(T(c["yt"]), T(c["yp"])).item()
    assert abs(got - c["expect"]) < 1e-6, (got, c["expect"])


def test_metrics_v3_kat():
    c = K.metrics_v3()
    t, p, g = T(c["yt"]), T(c["yp"]), (c["gh"], c["gw"])
    e = c["expect"]
    assert OM.obj_acc(t, p, g, c["A"], c["C"]).reshape(-1).tolist() == e["obj_acc"]
    assert abs(OM.mean_iou(t, p, g, c["A"], c["C"]).item() - e["mean_iou"]) < 1e-9
    assert abs(OM.class_acc(t, p, g, c["A"], c["C"]).item() - e["class_acc"]) < 1e-9
    assert abs(OM.recall(t, p, g, c["A"], c["C"], 0.5).item() - e["recall"]) < 1e-9


def test_metrics_v1_kat():
    c = K.metrics_v1()
    t, p, g = T(c["yt"]), T(c["yp"]), (c["gh"], c["gw"])
    e = c["expect"]
    assert OM.obj_acc_v1(t, p, g, c["B"], c["C"]).reshape(-1).tolist() == e["obj_acc"]
    assert abs(OM.mean_iou_v1(t, p, g, c["B"], c["C"]).item() - e["mean_iou"]) < 1e-9
    assert abs(OM.class_acc_v1(t, p, g, c["C"]).item() - e["class_acc"]) < 1e-9
    assert abs(OM.recall_v1(t, p, g, c["B"], c["C"], 0.5).item() - e["recall"]) < 1e-9


def test_keras_same_pad_rule():
    # (size, k, s) -> (out, pad_before): the numbers the reference's layers hit
    assert K.keras_same_pad(224, 7, 2) == (112, 2)      # v1.5 stem: total 5 -> 2 before, 3 after
    assert K.keras_same_pad(7, 3, 2) == (4, 1)          # v1.5 last down-sampling: 7 -> 4
    assert K.keras_same_pad(224, 3, 2) == (112, 0)      # even size: the single pad row goes AFTER
    assert K.keras_same_pad(13, 13, 1) == (13, 6)       # SPP 13x13 pool
    assert K.keras_same_pad(13, 2, 1) == (13, 0)        # tiny-YOLOv3 MaxPool(2, stride 1): pad after only
    for size, k, s in [(224, 7, 2), (7, 3, 2), (224, 3, 2), (13, 13, 1), (13, 2, 1), (8, 3, 2), (28, 7, 2)]:
        o, pb, pa = L.same_pad(size, k, s)
        assert (o, pb) == K.keras_same_pad(size, k, s) and pa >= pb


@pytest.mark.parametrize("c", K.conv_tap_cases())
def test_conv_padding_kat(c):
    k = c["k"]
    x = torch.zeros(1, c["H"], c["W"], 2, dtype=torch.float64)
    x[0, :, :, 0] = T(c["x"])
    w = torch.zeros(k, k, 2, 3, dtype=torch.float64)
    w[c["tap"][0], c["tap"][1], 0, 1] = 1.0
    y = L.conv2d(x, w, None, stride=c["stride"], padding=c["padding"])
    assert y.shape[1:3] == c["y"].shape
    assert torch.equal(y[0, :, :, 1], T(c["y"])) and float(y[0, :, :, 0].abs().max()) == 0.0


def test_space_to_depth_kat():
    x, y = K.space_to_depth_case()
    assert np.array_equal(L.space_to_depth2(T(x)).numpy(), y)


@pytest.mark.parametrize("c", K.maxpool_cases())
def test_maxpool_pad_kat(c):
    x = T(c["x"]).reshape(1, c["H"], c["W"], 1)
    y = L.maxpool(x, c["k"], c["stride"], "same")
    assert np.array_equal(y[0, :, :, 0].numpy(), c["y"])
