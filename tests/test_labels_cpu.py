"""Host label utilities (tf2_yolo_amd/labels.py) vs the reference's golden vectors and the oracle."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import gen_inputs  # noqa: E402

from oracle import tools as T  # noqa: E402
from tf2_yolo_amd import labels  # noqa: E402

G = np.load(os.path.join(HERE, "golden", "tools_golden.npz"))


def test_down2xlabel_and_class_weight_match_reference():
    m = gen_inputs.misc_inputs()
    l26 = labels.down2xlabel(m["label52"])
    assert np.array_equal(l26, G["label26"])
    assert np.array_equal(labels.down2xlabel(l26), G["label13"])
    pyr = labels.label_pyramid(m["label52"], 3)
    assert [p.shape[1] for p in pyr] == [13, 26, 52] and np.array_equal(pyr[0], G["label13"])
    assert np.array_equal(labels.get_class_weight(m["label52"][..., 4:5], "binary"), G["binary_weight52"])
    for meth in ("alpha", "log", "effective"):
        assert np.allclose(labels.get_class_weight(m["label52"][..., 5:], meth), G[f"class_weight_{meth}"],
                           rtol=1e-15, atol=0)


def test_encode_and_pyramid_match_oracle():
    rng = np.random.default_rng(3)
    boxes = rng.random((6, 4)) * 200
    boxes[:, 2:] = boxes[:, :2] + rng.random((6, 2)) * 150 + 5
    labs = rng.integers(0, 4, 6)
    a = labels.encode_boxes(boxes, labs, (416, 416), (52, 52), 4)
    b = T.encode_boxes(boxes, labs, (416, 416), (52, 52), 4)
    assert np.array_equal(a, b)
    even = rng.random((2, 8, 6, 6))
    even[..., 4] = (rng.random((2, 8, 6)) < 0.3)
    assert np.array_equal(labels.down2xlabel(even), T.down2xlabel(even))


def test_synthetic_batch_shapes():
    x, ys = labels.synthetic_batch(np.random.default_rng(0), 2, (64, 64), 5)
    assert x.shape == (2, 64, 64, 3) and x.dtype == np.float32
    assert [y.shape for y in ys] == [(2, 2, 2, 10), (2, 4, 4, 10), (2, 8, 8, 10)]
    assert ys[2][..., 4].sum() >= 2
