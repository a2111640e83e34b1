"""Pins the NumPy oracle (oracle/tools.py) to outputs of the reference's own code
(tests/golden/tools_golden.npz, produced by tests/golden/make_golden.py in the build container).
Bit-exact: these are integer/index selections and float64 row values."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import gen_inputs  # noqa: E402

from oracle import tools as T  # noqa: E402

G = np.load(os.path.join(HERE, "golden", "tools_golden.npz"))
CASES = list(gen_inputs.decode_cases())


@pytest.mark.parametrize("key,C,thr,lv", CASES, ids=[c[0] for c in CASES])
def test_decode_and_nms_match_reference(key, C, thr, lv):
    dec = T.decode(*lv, class_num=C, threshold=thr, version=3)
    ref = G[f"{key}_decode"]
    assert dec.shape == ref.shape and np.array_equal(dec, ref)
    assert np.array_equal(T.nms(dec, class_num=C, nms_threshold=0.5), G[f"{key}_nms"])
    assert np.array_equal(T.nms(dec, class_num=C, nms_threshold=0.5, iou_mode=2), G[f"{key}_diou"])
    assert np.array_equal(T.soft_nms(dec, class_num=C, nms_threshold=0.5, conf_threshold=thr, sigma=0.5),
                          G[f"{key}_soft"])


def test_decode_v1_v2_and_float64_labels():
    m = gen_inputs.misc_inputs()
    assert np.array_equal(T.decode(m["v1_lv"], class_num=4, threshold=0.4, version=1), G["v1_decode"])
    assert np.array_equal(T.decode(m["v2_lv"], class_num=20, threshold=0.8, version=2), G["v2_decode"])
    assert np.array_equal(T.decode(m["label52"][0], class_num=3, threshold=0.5, version=3), G["label52_decode"])
    with pytest.raises(ValueError, match="Invalid version"):
        T.decode(m["v1_lv"], class_num=4, version=7)


def test_label_pyramid_and_class_weights():
    m = gen_inputs.misc_inputs()
    l26 = T.down2xlabel(m["label52"])
    assert np.array_equal(l26, G["label26"])
    assert np.array_equal(T.down2xlabel(l26), G["label13"])
    assert np.array_equal(T.get_class_weight(m["label52"][..., 4:5], "binary"), G["binary_weight52"])
    for meth in ("alpha", "log", "effective"):
        assert np.array_equal(T.get_class_weight(m["label52"][..., 5:], meth), G[f"class_weight_{meth}"])


def test_pairwise_iou():
    m = gen_inputs.misc_inputs()
    b = m["iou_boxes"]
    assert np.array_equal(T.cal_iou(b.reshape(-1, 1, 5), b.reshape(1, -1, 5), mode=1), G["iou_mat"])
    assert np.array_equal(T.cal_iou(b.reshape(-1, 1, 5), b.reshape(1, -1, 5), mode=2), G["diou_mat"])


def test_empty_and_out_of_range_classes():
    empty = np.zeros((0, 7))
    assert T.nms(empty, class_num=3).shape == (0, 7)
    rows = np.array([[.5, .5, .2, .2, .9, 5., .9], [.5, .5, .2, .2, .8, 0., .9]])
    out = T.nms(rows, class_num=2)          # class 5 >= class_num is dropped by the per-class gather
    assert out.shape == (1, 7) and out[0, 5] == 0
