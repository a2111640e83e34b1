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


def measurement_inputs(seed=5, n_img=8, C=3, A=3, g=8):
    """Evaluation fixture (utils/measurement.py): labels on a g x g grid (float64, as the data path makes them)
    and two prediction levels (g and g/2, float32) that contain noisy copies of the labelled boxes (true
    positives of varying quality, some with the wrong class, some duplicated) plus random false positives.
    Scores are continuous random numbers: no ties."""
    rng = np.random.default_rng(seed)
    y_true = np.zeros((n_img, g, g, 5 + C))
    lv0 = np.zeros((n_img, g, g, A * (5 + C)), dtype=np.float32)
    lv1 = np.zeros((n_img, g // 2, g // 2, A * (5 + C)), dtype=np.float32)
    v0 = lv0.reshape(n_img, g, g, A, 5 + C)
    v1 = lv1.reshape(n_img, g // 2, g // 2, A, 5 + C)
    for b in range(n_img):
        for _ in range(int(rng.integers(3, 9))):
            y, x = rng.integers(0, g, 2)
            if y_true[b, y, x, 4] == 1:
                continue
            xy = rng.random(2)
            wh = rng.random(2) * 0.3 + 0.08
            c = int(rng.integers(0, C))
            y_true[b, y, x, :2] = xy
            y_true[b, y, x, 2:4] = wh
            y_true[b, y, x, 4] = 1
            y_true[b, y, x, 5 + c] = 1
            for a in range(A):   # up to A noisy copies on the fine level
                if rng.random() < 0.6:
                    noise = rng.normal(0, 0.08 * (1 + a), 4)
                    v0[b, y, x, a, :2] = np.clip(xy + noise[:2] * 0.5, 0.01, 0.99)
                    v0[b, y, x, a, 2:4] = np.clip(wh * (1 + noise[2:]), 0.02, 0.9)
                    v0[b, y, x, a, 4] = rng.random() * 0.7 + 0.3
                    cc = c if rng.random() < 0.8 else int(rng.integers(0, C))
                    v0[b, y, x, a, 5:] = rng.random(C) * 0.2
                    v0[b, y, x, a, 5 + cc] = rng.random() * 0.5 + 0.5
            if rng.random() < 0.5:   # a coarse-level copy (cell = (y//2, x//2), offset re-expressed)
                a = int(rng.integers(0, A))
                v1[b, y // 2, x // 2, a, 0] = ((x + xy[0]) / 2) % 1
                v1[b, y // 2, x // 2, a, 1] = ((y + xy[1]) / 2) % 1
                v1[b, y // 2, x // 2, a, 2:4] = wh * (1 + rng.normal(0, 0.1, 2))
                v1[b, y // 2, x // 2, a, 4] = rng.random() * 0.6 + 0.2
                v1[b, y // 2, x // 2, a, 5:] = rng.random(C) * 0.3
                v1[b, y // 2, x // 2, a, 5 + c] = rng.random() * 0.5 + 0.5
        for _ in range(int(rng.integers(4, 12))):   # false positives
            y, x, a = rng.integers(0, g), rng.integers(0, g), rng.integers(0, A)
            if v0[b, y, x, a, 4] > 0:
                continue
            v0[b, y, x, a, :2] = rng.random(2)
            v0[b, y, x, a, 2:4] = rng.random(2) * 0.3 + 0.05
            v0[b, y, x, a, 4] = rng.random() * 0.8 + 0.1
            v0[b, y, x, a, 5:] = rng.random(C) * 0.9 + 0.05
    return y_true, lv0, lv1
