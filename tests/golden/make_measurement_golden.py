"""Golden vectors for the evaluation path, made by EXECUTING THE REFERENCE's own utils/measurement.py
(build container only; same stub modules as make_golden.py). Only outputs are stored
(tests/golden/measurement_golden.npz); the inputs are regenerated from gen_inputs.measurement_inputs().

Run:  python -B tests/golden/make_measurement_golden.py
"""
import os
import sys
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_inputs                      # noqa: E402
from make_golden import import_reference_tools   # noqa: E402

SCORE_CASES = [  # (key, kwargs of create_score_mat)
    ("s0", dict(conf_threshold=0.5, nms_mode=0, precision_mode=2)),
    ("s1", dict(conf_threshold=0.3, nms_mode=1, nms_threshold=0.5, precision_mode=0)),
    ("s2", dict(conf_threshold=0.3, nms_mode=3, nms_threshold=0.4, precision_mode=1, iou_threshold=0.4)),
    ("s3", dict(conf_threshold=0.25, nms_mode=2, nms_threshold=0.5, nms_sigma=0.5, precision_mode=2, iou_threshold=0.6)),
]
PR_CASES = [  # (key, kwargs of PRfunc)
    ("p0", dict(conf_threshold=0.05, nms_mode=1, precision_mode=2, max_per_img=100)),
    ("p1", dict(conf_threshold=0.1, nms_mode=0, precision_mode=0, max_per_img=5)),
    ("p2", dict(conf_threshold=0.05, nms_mode=3, nms_threshold=0.45, precision_mode=1, max_per_img=None, iou_threshold=0.4)),
]
CLASS_NAMES = ["a", "b", "c"]


def main():
    import_reference_tools()
    from utils import measurement as M   # the reference's utils/measurement.py
    y_true, lv0, lv1 = gen_inputs.measurement_inputs()
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for key, kw in SCORE_CASES:
            t = M.create_score_mat(y_true, lv0, lv1, class_names=CLASS_NAMES, version=3, **kw)
            for col in ("precision", "recall", "F1-score", "gts", "dets"):
                out[f"{key}_{col}"] = t[col].to_numpy()
        for key, kw in PR_CASES:
            f = M.PRfunc(y_true, lv0, lv1, class_names=CLASS_NAMES, version=3, **kw)
            for c in range(len(CLASS_NAMES)):
                assert len(f.precisions[c]) > 1, "class without detections: pick another seed"
                out[f"{key}_prec{c}"] = np.asarray(f.precisions[c], dtype=np.float64)
                out[f"{key}_rec{c}"] = np.asarray(f.recalls[c], dtype=np.float64)
            for mode in ("voc2007", "voc2012", "area", "smootharea"):
                out[f"{key}_map_{mode}"] = f.get_map(mode)["ap"].to_numpy().astype(np.float64)
            out[f"{key}_call"] = np.array([[f(r, c) for r in (0.0, 0.3, 0.55, 0.9)] for c in range(3)], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "measurement_golden.npz"), **out)
    print("arrays:", len(out), {k: v.shape for k, v in list(out.items())[:6]})


if __name__ == "__main__":
    main()
