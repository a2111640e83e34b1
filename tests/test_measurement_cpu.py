"""The NumPy oracle of the evaluation path (oracle/measurement.py) against outputs of the reference's own
utils/measurement.py (tests/golden/measurement_golden.npz, made by tests/golden/make_measurement_golden.py).
Bit-exact: the curves are ratios of integer counts."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import gen_inputs                                    # noqa: E402
from make_measurement_golden import PR_CASES, SCORE_CASES   # noqa: E402

from oracle import measurement as OM                  # noqa: E402

GOLD = np.load(os.path.join(HERE, "golden", "measurement_golden.npz"))


def _eq(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("key,kw", SCORE_CASES)
def test_oracle_score_table_matches_reference(key, kw):
    y_true, lv0, lv1 = gen_inputs.measurement_inputs()
    kw = dict(kw)
    pm = kw.pop("precision_mode")
    counts = OM.score_counts(y_true, (lv0, lv1), 3, version=3, **kw)
    p, r, f1 = OM.score_table(counts, pm)
    assert _eq(p, GOLD[f"{key}_precision"]) and _eq(r, GOLD[f"{key}_recall"]) and _eq(f1, GOLD[f"{key}_F1-score"])
    assert np.array_equal(counts[:, 1], GOLD[f"{key}_gts"]) and np.array_equal(counts[:, 0], GOLD[f"{key}_dets"])


@pytest.mark.parametrize("key,kw", PR_CASES)
def test_oracle_pr_curves_match_reference(key, kw):
    y_true, lv0, lv1 = gen_inputs.measurement_inputs()
    ps, rs = OM.pr_curves(y_true, (lv0, lv1), 3, version=3, **kw)
    for c in range(3):
        assert _eq(ps[c], GOLD[f"{key}_prec{c}"]), (key, c)
        assert _eq(rs[c], GOLD[f"{key}_rec{c}"]), (key, c)
    for mode in ("voc2007", "voc2012", "area", "smootharea"):
        assert _eq(OM.average_precisions(ps, rs, mode), GOLD[f"{key}_map_{mode}"]), (key, mode)
    calls = [[OM.precision_at(ps[c], rs[c], r) for r in (0.0, 0.3, 0.55, 0.9)] for c in range(3)]
    assert _eq(calls, GOLD[f"{key}_call"])
