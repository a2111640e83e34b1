"""Deterministic inputs shared by make_golden.py (reference side) and the tests (our side).
numpy.random.default_rng (PCG64) streams are stable across NumPy versions, so only seeds and the
reference's OUTPUTS are stored in tools_golden.npz."""
import numpy as np


def levels(rng, grids, A, C, sparse):
    out = []
    for g in grids:
        a = rng.random((g, g, A * (5 + C)), dtype=np.float32)
        if sparse:
            a = a ** 6
            a.reshape(g, g, A, 5 + C)[..., 2:4] = rng.random((g, g, A, 2), dtype=np.float32) * 0.5 + 0.02
        out.append(a)
    return out


def decode_cases():
    """yields (key, C, thr, level arrays fine->coarse)"""
    case = 0
    for seed in (0, 1, 2):
        for C in (1, 3, 80):
            for thr in (0.5, 0.9):
                rng = np.random.default_rng(1000 * seed + 10 * C + int(thr * 10))
                grids = (13, 26) if C == 80 else (13, 26, 52)
                lv = levels(rng, grids[::-1], 3, C, sparse=(thr == 0.5))
                yield f"c{case}", C, thr, lv
                case += 1


def misc_inputs():
    rng = np.random.default_rng(77)
    d = {}
    d["v1_lv"] = rng.random((7, 7, 2 * 5 + 4), dtype=np.float32)
    d["v2_lv"] = rng.random((13, 13, 5 * (5 + 20)), dtype=np.float32)
    lab = np.zeros((2, 52, 52, 5 + 3))
    for b in range(2):
        for _ in range(12):
            y, x = rng.integers(0, 52, 2)
            lab[b, y, x, :2] = rng.random(2)
            lab[b, y, x, 2:4] = rng.random(2) * 0.5 + 0.05
            lab[b, y, x, 4] = 1
            lab[b, y, x, 5 + rng.integers(0, 3)] = 1
    d["label52"] = lab
    d["iou_boxes"] = rng.random((40, 5))
    return d
