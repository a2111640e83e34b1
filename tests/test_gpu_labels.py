"""Label tensors built on the device (csrc/labels.hip: yolo_encode_labels, yolo_down2xlabel) -- SURVEY.md section 8f
row 2 -- bit for bit against the reference's own outputs (tests/golden/tools_golden.npz: label26 / label13 are
utils.tools.down2xlabel run on the seeded label52) and against the oracle's restatement of the encoder
(utils/tools.py:179-209) on boxes that collide in a cell, touch the right / bottom edge and start outside the image."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import gen_inputs  # noqa: E402

from oracle import tools as T  # noqa: E402

pytestmark = pytest.mark.gpu
G = np.load(os.path.join(HERE, "golden", "tools_golden.npz"))


def test_down2xlabel_matches_the_reference_bit_for_bit():
    from tf2_yolo_amd import ops
    l52 = gen_inputs.misc_inputs()["label52"]
    d64, d32 = ops.down2xlabel(torch.from_numpy(np.ascontiguousarray(l52)).cuda())
    assert np.array_equal(d64.cpu().numpy(), G["label26"])
    assert np.array_equal(d32.cpu().numpy(), G["label26"].astype(np.float32))
    e64, e32 = ops.down2xlabel(d64)
    assert np.array_equal(e64.cpu().numpy(), G["label13"])
    assert np.array_equal(e32.cpu().numpy(), G["label13"].astype(np.float32))
    # odd grids: the last row / column is dropped like the reference's range(0, g, 2) with g // 2 outputs ... only
    # even grids occur in the reference's use (52 -> 26 -> 13); an odd one must at least agree with the oracle
    rng = np.random.default_rng(5)
    lab = rng.random((2, 6, 10, 7))
    lab[..., 4] = rng.random((2, 6, 10)) < 0.3
    lab[lab[..., 4] == 0] = 0
    o64, _ = ops.down2xlabel(torch.from_numpy(lab).cuda())
    assert np.array_equal(o64.cpu().numpy(), T.down2xlabel(lab))


def test_encoder_and_pyramid_match_the_oracle_bit_for_bit():
    from tf2_yolo_amd import labels
    rng = np.random.default_rng(11)
    H = W = 416
    C = 5
    boxes, classes = [], []
    for n in range(6):
        k = int(rng.integers(0, 9))                    # image 0..: also an image with no box at all
        c = rng.random((k, 2)) * [W, H]
        wh = rng.uniform(4, 250, (k, 2))
        b = np.concatenate([c - wh / 2, c + wh / 2], axis=1)
        if n == 1 and k:
            b[0] = [W - 3.0, H - 2.5, W + 5.0, H + 1.5]     # centre beyond the right / bottom edge: skipped (:197)
        if n == 2 and k > 1:
            b[1] = b[0] + 0.25                              # same cell as box 0: last writer wins, class bits accumulate
        if n == 3 and k:
            b[0] = [-30.0, 10.0, 10.0, 50.0]                # centre left of the image: NumPy's negative index wraps
        boxes.append(b)
        classes.append(rng.integers(0, C, k))
    got = labels.label_pyramid_device(boxes, classes, (H, W), (52, 52), C, 3)
    fine = np.stack([T.encode_boxes(b, c, (H, W), (52, 52), C) for b, c in zip(boxes, classes)])
    ref = [fine]
    for _ in range(2):
        ref.insert(0, T.down2xlabel(ref[0]))
    assert [tuple(g.shape) for g in got] == [(6, 13, 13, 5 + C), (6, 26, 26, 5 + C), (6, 52, 52, 5 + C)]
    for g, r in zip(got, ref):
        assert g.dtype == torch.float32 and np.array_equal(g.cpu().numpy(), r.astype(np.float32))
    # and the host implementation the training loop used so far gives the same tensors
    host = labels.label_pyramid(np.stack([labels.encode_boxes(b, c, (H, W), (52, 52), C) for b, c in zip(boxes, classes)]), 3)
    for g, r in zip(got, host):
        assert np.array_equal(g.cpu().numpy(), r.astype(np.float32))
