"""GPU evaluation path (utils.measurement drop-in: create_score_mat, PRfunc) against the outputs of the
reference's own utils/measurement.py (golden) and against the NumPy oracle on further inputs. Bit-exact."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import gen_inputs                                    # noqa: E402
from make_measurement_golden import CLASS_NAMES, PR_CASES, SCORE_CASES   # noqa: E402

from oracle import measurement as OM                  # noqa: E402

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(HERE, "golden", "measurement_golden.npz"))


def _eq(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("key,kw", SCORE_CASES)
def test_create_score_mat_matches_reference(key, kw):
    from utils.measurement import create_score_mat
    y_true, lv0, lv1 = gen_inputs.measurement_inputs()
    t = create_score_mat(y_true, lv0, lv1, class_names=CLASS_NAMES, version=3, **kw)
    assert list(t.columns) == ["precision", "recall", "F1-score", "gts", "dets"] and list(t.index) == CLASS_NAMES
    for col in ("precision", "recall", "F1-score", "gts", "dets"):
        assert _eq(t[col].to_numpy(), GOLD[f"{key}_{col}"]), (key, col)


@pytest.mark.parametrize("key,kw", PR_CASES)
def test_prfunc_matches_reference(key, kw):
    from utils.measurement import PRfunc
    y_true, lv0, lv1 = gen_inputs.measurement_inputs()
    f = PRfunc(y_true, lv0, lv1, class_names=CLASS_NAMES, version=3, **kw)
    for c in range(3):
        assert _eq(f.precisions[c], GOLD[f"{key}_prec{c}"]), (key, c)
        assert _eq(f.recalls[c], GOLD[f"{key}_rec{c}"]), (key, c)
    for mode in ("voc2007", "voc2012", "area", "smootharea"):
        m = f.get_map(mode)
        assert list(m.index) == CLASS_NAMES + ["mAP"] and list(m.columns) == ["ap"]
        assert _eq(m["ap"].to_numpy(), GOLD[f"{key}_map_{mode}"]), (key, mode)
    calls = [[f(r, c) for r in (0.0, 0.3, 0.55, 0.9)] for c in range(3)]
    assert _eq(calls, GOLD[f"{key}_call"])
    with pytest.raises(IndexError):
        f(0.5, 3)


def test_larger_set_against_oracle():
    """more images / classes than the golden fixture, soft-NMS, a tight per-image cap"""
    from utils.measurement import PRfunc, create_score_mat
    y_true, lv0, lv1 = gen_inputs.measurement_inputs(seed=21, n_img=40, C=5, A=3, g=8)
    names = [f"k{i}" for i in range(5)]
    kw = dict(conf_threshold=0.2, nms_mode=2, nms_threshold=0.5, nms_sigma=0.4, iou_threshold=0.5)
    t = create_score_mat(y_true, lv0, lv1, class_names=names, precision_mode=1, version=3, **kw)
    counts = OM.score_counts(y_true, (lv0, lv1), 5, version=3, **kw)
    p, r, f1 = OM.score_table(counts, 1)
    assert _eq(t["precision"].to_numpy(), p) and _eq(t["recall"].to_numpy(), r) and _eq(t["F1-score"].to_numpy(), f1)
    f = PRfunc(y_true, lv0, lv1, class_names=names, precision_mode=2, max_per_img=3, version=3, **kw)
    ps, rs = OM.pr_curves(y_true, (lv0, lv1), 5, precision_mode=2, max_per_img=3, version=3, **kw)
    for c in range(5):
        assert _eq(f.precisions[c], ps[c]) and _eq(f.recalls[c], rs[c]), c
    assert _eq(f.get_map("area")["ap"].to_numpy(), OM.average_precisions(ps, rs, "area"))


def test_empty_corners():
    """no detections at all / a class nobody predicts: defined results instead of the reference's NameError"""
    from utils.measurement import PRfunc, create_score_mat
    y_true, lv0, lv1 = gen_inputs.measurement_inputs(seed=3, n_img=2, C=3)
    t = create_score_mat(y_true, lv0 * 0, lv1 * 0, class_names=["a", "b", "c"], version=3)
    assert (t["dets"].to_numpy() == 0).all() and np.isnan(t["precision"].to_numpy()).all()
    f = PRfunc(y_true, lv0 * 0, lv1 * 0, class_names=["a", "b", "c"], version=3)
    assert all(len(p) == 1 and p[0] == 0 for p in f.precisions)
    assert f.get_map("voc2007")["ap"].to_numpy()[-1] == 0


def test_detection_writers_match_reference_files(tmp_path):
    """array_to_json / array_to_xml: byte-identical files (the json after undoing the `np.float64(...)`
    wrappers that the reference's str(dict) leaks under NumPy 2 -- the numbers inside are identical)"""
    import json
    import re
    from make_writers_golden import CASES, IMG_SIZE, NAMES
    from utils.tools import array_to_json, array_to_xml
    gold = np.load(os.path.join(HERE, "golden", "writers_golden.npz"))
    _, lv0, lv1 = gen_inputs.measurement_inputs()
    for key, img, kw in CASES:
        pj, px = str(tmp_path / (key + ".json")), str(tmp_path / (key + ".xml"))
        array_to_json(pj, IMG_SIZE, lv0[img], lv1[img], class_names=NAMES, version=3, **kw)
        array_to_xml(px, IMG_SIZE, lv0[img], lv1[img], class_names=NAMES, version=3, **kw)
        want_json = re.sub(rb"np\.float64\(([^)]*)\)", rb"\1", gold[key + "_json"].tobytes())
        assert open(pj, "rb").read() == want_json, key
        assert open(px, "rb").read() == gold[key + "_xml"].tobytes(), key
        json.loads(open(pj, encoding="big5").read())   # and it is json
