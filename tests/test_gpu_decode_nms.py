"""GPU decode / NMS vs the reference's golden vectors (bit-exact) and vs the oracle on larger,
denser inputs (bit-exact index selection; BASELINE.json north_star)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import gen_inputs  # noqa: E402

from oracle import tools as T  # noqa: E402

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(HERE, "golden", "tools_golden.npz"))
CASES = list(gen_inputs.decode_cases())


@pytest.mark.parametrize("key,C,thr,lv", CASES, ids=[c[0] for c in CASES])
def test_golden(key, C, thr, lv):
    from tf2_yolo_amd import tools
    dec = tools.decode(*lv, class_num=C, threshold=thr, version=3)
    assert np.array_equal(dec.reshape(-1, 7), G[f"{key}_decode"])
    assert np.array_equal(tools.nms(dec, class_num=C, nms_threshold=0.5), G[f"{key}_nms"])
    assert np.array_equal(tools.nms(dec, class_num=C, nms_threshold=0.5, iou_mode=2), G[f"{key}_diou"])
    assert np.array_equal(tools.soft_nms(dec, class_num=C, nms_threshold=0.5, conf_threshold=thr, sigma=0.5),
                          G[f"{key}_soft"])


def test_golden_v1_v2_f64():
    from tf2_yolo_amd import tools
    m = gen_inputs.misc_inputs()
    assert np.array_equal(tools.decode(m["v1_lv"], class_num=4, threshold=0.4, version=1), G["v1_decode"])
    assert np.array_equal(tools.decode(m["v2_lv"], class_num=20, threshold=0.8, version=2), G["v2_decode"])
    assert np.array_equal(tools.decode(m["label52"][0], class_num=3, threshold=0.5, version=3), G["label52_decode"])
    with pytest.raises(ValueError, match="Invalid version"):
        tools.decode(m["v1_lv"], class_num=4, version=7)


def test_dense_vs_oracle():
    """BASELINE config 5 shape: YOLOv3-416, C=80, uniform noise, thr .9 (4 425 candidates)."""
    from tf2_yolo_amd import tools
    rng = np.random.default_rng(1234)
    lv = [rng.random((g, g, 255), dtype=np.float32) for g in (52, 26, 13)]
    dec = tools.decode(*lv, class_num=80, threshold=0.9, version=3)
    ref = T.decode(*lv, class_num=80, threshold=0.9, version=3)
    assert np.array_equal(dec, ref)
    assert np.array_equal(tools.nms(dec, class_num=80, nms_threshold=0.5), T.nms(ref, 80, 0.5))
    assert np.array_equal(tools.nms(dec, class_num=80, nms_threshold=0.5, iou_mode=2), T.nms(ref, 80, 0.5, 2))
    assert np.array_equal(tools.soft_nms(dec, class_num=80, nms_threshold=0.5, conf_threshold=0.9, sigma=0.5),
                          T.soft_nms(ref, 80, 0.5, 0.9, 0.5))


def test_single_class_many_boxes_and_properties():
    from tf2_yolo_amd import tools
    rng = np.random.default_rng(5)
    n = 3000
    rows = np.zeros((n, 7))
    rows[:, :2] = rng.random((n, 2))
    rows[:, 2:4] = rng.random((n, 2)) * 0.3 + 0.02
    rows[:, 4] = rng.random(n)
    rows[:, 6] = rng.random(n)
    out = tools.nms(rows, class_num=1, nms_threshold=0.3)
    assert np.array_equal(out, T.nms(rows, 1, 0.3))
    # idempotence and subset-in-order
    assert np.array_equal(tools.nms(out, class_num=1, nms_threshold=0.3), out)
    # empty input
    assert tools.nms(np.zeros((0, 7)), class_num=3).shape == (0, 7)
    # out-of-range class ids are dropped like the reference's per-class gather does
    rows2 = np.array([[.5, .5, .2, .2, .9, 5., .9], [.5, .5, .2, .2, .8, 0., .9]])
    out2 = tools.nms(rows2, class_num=2)
    assert out2.shape == (1, 7) and out2[0, 5] == 0


def test_decode_capacity_regrow():
    from tf2_yolo_amd import tools
    rng = np.random.default_rng(9)
    lv = rng.random((26, 26, 255), dtype=np.float32)
    full = T.decode(lv, class_num=80, threshold=0.3, version=3)
    got = tools.decode_device(lv, class_num=80, threshold=0.3, version=3, capacity=100).cpu().numpy()
    assert np.array_equal(got, full)
