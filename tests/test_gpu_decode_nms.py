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


def _crowded_rows(rng, n, class_num, skew=False):
    """rows whose boxes overlap heavily (centres in a small region, similar sizes): most of them get suppressed, the
    greedy walk's decisions depend on one another in long chains"""
    rows = np.zeros((n, 7))
    rows[:, :2] = 0.3 + 0.4 * rng.random((n, 2))
    rows[:, 2:4] = 0.05 + 0.1 * rng.random((n, 2))
    rows[:, 4] = rng.random(n)
    rows[:, 6] = rng.random(n)
    if skew:     # a few large classes and many small ones
        p = np.ones(class_num)
        p[:3] = class_num * 2.0
        rows[:, 5] = rng.choice(class_num, size=n, p=p / p.sum())
    else:
        rows[:, 5] = rng.integers(0, class_num, n)
    return rows


@pytest.mark.parametrize("case", ["crowded_20_classes", "skewed_80_classes", "one_class_5000", "uniform_noise_131k",
                                  "class_larger_than_the_bit_matrix"])
def test_nms_bit_matrix_and_walk_give_the_same_rows(case):
    """Round 5: hard / DIoU NMS makes every pair test of a class first (bit matrix, whole chip) and then walks the bits
    (csrc/decode_nms.hip: nms_mask_kernel, nms_scan_kernel); classes of more than 8192 rows keep the greedy walk kernel
    (nms_walk_kernel), which yolo_set_option(7, 1) selects for every class. Both forms must return bit-identical rows --
    crowded inputs where most rows are suppressed in long dependency chains, skewed class sizes (tile rows of different
    lengths, classes of one row), a class that spans 79 tile rows, BASELINE.md's 131 304-row uniform-noise input -- and on the
    inputs the CPU oracle can do in seconds both must equal the oracle's rows."""
    import torch
    from tf2_yolo_amd import ops, tools
    rng = np.random.default_rng(77)
    check_oracle = True
    if case == "crowded_20_classes":
        rows, C = _crowded_rows(rng, 6000, 20), 20
    elif case == "skewed_80_classes":
        rows, C = _crowded_rows(rng, 9000, 80, skew=True), 80
    elif case == "one_class_5000":
        rows, C = _crowded_rows(rng, 5000, 1), 1
    elif case == "class_larger_than_the_bit_matrix":
        rows, C = _crowded_rows(rng, 9500, 2), 2
        rows[:9000, 5] = 1            # class 1: 9000 rows (> 8192: walk kernel in the default mode too), class 0: the rest
        rows[9000:, 5] = 0
        rows[:, :2] = rng.random((9500, 2))     # spread out: a few thousand survivors
        check_oracle = False          # (a 9000 x 9000 float64 IoU matrix on the host)
    else:
        lv = [torch.from_numpy(np.random.default_rng(1234).random((g, g, 255), dtype=np.float32)).cuda() for g in (13, 26, 52)]
        rows, C = tools.decode_device(*lv, class_num=80, threshold=0.5, version=3), 80
        assert rows.shape[0] > 100000
        check_oracle = False
    dev = rows if torch.is_tensor(rows) else torch.from_numpy(rows).cuda()
    for thr in (0.5, 0.2):
        got = {}
        try:
            for walk in (0, 1):
                ops.set_option(ops.OPT_NMS_WALK, walk)
                got[walk] = (tools.nms(dev, class_num=C, nms_threshold=thr), tools.nms(dev, class_num=C, nms_threshold=thr, iou_mode=2))
        finally:
            ops.reset_options()
        for a, b in zip(got[0], got[1]):
            assert torch.equal(a, b)
        assert got[0][0].shape[0] < dev.shape[0] or case == "uniform_noise_131k"
        if check_oracle:
            assert np.array_equal(got[0][0].cpu().numpy(), T.nms(rows, C, thr))
            assert np.array_equal(got[0][1].cpu().numpy(), T.nms(rows, C, thr, 2))
    # a threshold of zero or below: no early rejection of non-overlapping pairs (IoU = 0 >= 0 suppresses everything behind
    # the best row of a class; DIoU is negative for distinct centres)
    small = dev[:1500].contiguous()
    for thr in (0.0, -0.5):
        for mode in (1, 2):
            ref = T.nms(small.cpu().numpy(), C, thr, mode)
            assert np.array_equal(tools.nms(small, class_num=C, nms_threshold=thr, iou_mode=mode).cpu().numpy(), ref)
    # soft-NMS (LDS-tiled since round 5) on the same rows against the oracle
    if check_oracle:
        assert np.array_equal(tools.soft_nms(dev, class_num=C, nms_threshold=0.3, conf_threshold=0.3, sigma=0.5).cpu().numpy(),
                              T.soft_nms(rows, C, 0.3, 0.3, 0.5))


def test_nms_edge_cases_walk_and_bit_matrix_agree():
    """The corners of the round-5 NMS pipeline: more classes than the LDS-resident counters hold (3000 > 2048: the global-atomic
    paths of the counter kernels and the tile table read from memory), inputs of 1 / 2 / 3 rows, exact score ties inside a
    class (ordered "higher original index first" in both forms), every row in one class with identical boxes (one survivor)."""
    import torch
    from tf2_yolo_amd import ops, tools
    rng = np.random.default_rng(99)

    def both(rows, C, thr=0.5):
        dev = torch.from_numpy(rows).cuda()
        out = {}
        try:
            for walk in (0, 1):
                ops.set_option(ops.OPT_NMS_WALK, walk)
                out[walk] = [tools.nms(dev, class_num=C, nms_threshold=thr).cpu().numpy(),
                             tools.nms(dev, class_num=C, nms_threshold=thr, iou_mode=2).cpu().numpy(),
                             tools.soft_nms(dev, class_num=C, nms_threshold=thr, conf_threshold=0.3, sigma=0.5).cpu().numpy()]
        finally:
            ops.reset_options()
        for a, b in zip(out[0], out[1]):
            assert np.array_equal(a, b)
        return out[0]

    rows = _crowded_rows(rng, 6000, 3000)
    got = both(rows, 3000)
    assert np.array_equal(got[0], T.nms(rows, 3000, 0.5)) and np.array_equal(got[1], T.nms(rows, 3000, 0.5, 2))
    assert np.array_equal(got[2], T.soft_nms(rows, 3000, 0.5, 0.3, 0.5))
    for n in (1, 2, 3):
        r = _crowded_rows(rng, n, 2)
        g = both(r, 2)
        assert np.array_equal(g[0], T.nms(r, 2, 0.5))
    # exact ties: groups of equal conf * prob inside one class
    r = _crowded_rows(rng, 400, 1)
    r[:, 4] = np.repeat(rng.random(40), 10)
    r[:, 6] = 0.5
    g = both(r, 1, thr=0.3)
    assert 0 < g[0].shape[0] < 400
    # identical boxes: exactly one survivor (the best-ranked row; ties -> the highest original index)
    r = np.tile(np.array([[.5, .5, .2, .2, .9, 0., .9]]), (300, 1))
    g = both(r, 1)
    assert g[0].shape[0] == 1 and g[1].shape[0] == 1


def test_nms_with_nan_scores_keeps_a_total_order():
    """yolo_nms is a raw C-ABI and takes any rows. A NaN score compares false both ways; the LDS bitonic
    sort needs a total order or its (-inf, -1) padding can end among the first nc entries (row index -1: an out-of-bounds
    access). The order key of a NaN score is +inf -- np.argsort puts NaN last, so the reference's argsort()[::-1]
    (utils/tools.py:717) visits NaN rows FIRST. Class sizes are not powers of two (padding present), one class exceeds the
    bit-matrix bound (rank by counting). Both forms agree bit for bit; with ONE NaN per class the order is defined and the
    result equals the oracle's."""
    import torch
    from tf2_yolo_amd import ops, tools
    rng = np.random.default_rng(5)

    def both(rows, C, thr=0.5, soft=True):
        dev = torch.from_numpy(rows).cuda()
        out = {}
        try:
            for walk in (0, 1):
                ops.set_option(ops.OPT_NMS_WALK, walk)
                out[walk] = [tools.nms(dev, class_num=C, nms_threshold=thr).cpu().numpy(),
                             tools.nms(dev, class_num=C, nms_threshold=thr, iou_mode=2).cpu().numpy()]
                if soft:
                    out[walk].append(tools.soft_nms(dev, class_num=C, nms_threshold=thr, conf_threshold=0.3, sigma=0.5).cpu().numpy())
        finally:
            ops.reset_options()
        for a, b in zip(out[0], out[1]):
            assert np.array_equal(a, b, equal_nan=True)
        return out[0]

    # one NaN per class, 7 classes of ~143 rows (N2 = 256 > nc): defined order, equal to the oracle
    rows = _crowded_rows(rng, 1000, 7)
    for c in range(7):
        idx = np.flatnonzero(rows[:, 5] == c)
        rows[idx[len(idx) // 2], 4] = np.nan
    g = both(rows, 7, soft=False)
    assert np.array_equal(g[0], T.nms(rows, 7, 0.5), equal_nan=True)
    assert np.array_equal(g[1], T.nms(rows, 7, 0.5, 2), equal_nan=True)
    assert np.isnan(g[0][:, 4]).sum() == 7          # the NaN row of every class is visited first and kept
    # many NaN rows (a third of a 3000-row class; N2 = 4096), +-inf scores among them: every returned row is an input row,
    # no row twice, both forms identical
    rows = _crowded_rows(rng, 3000, 1)
    rows[rng.choice(3000, 1000, replace=False), 6] = np.nan
    rows[5, 4], rows[6, 4] = np.inf, -np.inf
    for out in both(rows, 1):
        assert 0 < out.shape[0] <= 3000
        key = {r.tobytes() for r in rows}
        assert all(r.tobytes() in key for r in out)
    # a class beyond the bit-matrix bound (9000 rows: rank by counting + walk kernel) with NaN scores
    rows = _crowded_rows(rng, 9000, 1)
    rows[:, :2] = rng.random((9000, 2))
    rows[rng.choice(9000, 500, replace=False), 4] = np.nan
    out = both(rows, 1, soft=False)
    assert 0 < out[0].shape[0] <= 9000 and len({r.tobytes() for r in out[0]}) == out[0].shape[0]
