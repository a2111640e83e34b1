"""CPU timings of the REFERENCE's own decode / nms / soft_nms (utils/tools.py:370-438, :687-786) on the BASELINE.md
inputs, produced in the build container (the reference never travels to the GPU box) and committed as
tests/golden/tools_timing.json: the CPU side of BASELINE.json config 5, quoted beside the GPU numbers by bench.py.

Run:  python -B tests/golden/make_timing.py          (about 3 minutes; single-threaded Python by construction)
Inputs: ONE numpy.random.default_rng(1234) drawing float32 uniform levels (13,13,255), (26,26,255), (52,52,255) in
that order; C = 80; nms_threshold = 0.5, sigma = 0.5. The counts must be BASELINE.md's 131 304 / 4 425 candidates.
"""
import json
import os
import platform
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import import_reference_tools  # noqa: E402


def levels():
    rng = np.random.default_rng(1234)
    return [rng.random((g, g, 255), dtype=np.float32) for g in (13, 26, 52)]


def med(fn, n):
    ts = []
    out = None
    for _ in range(n):
        t0 = time.perf_counter()
        out = fn()
        ts.append(time.perf_counter() - t0)
    return float(np.median(ts)) * 1e3, out, n


def main():
    tools = import_reference_tools()
    lv = levels()
    res = {"what": "reference utils.tools decode / nms / soft_nms, CPU, 1 thread (pure Python + NumPy)",
           "host": {"cpu": platform.processor() or platform.machine(), "cpus_visible": os.cpu_count(),
                    "numpy": np.__version__, "python": platform.python_version()},
           "inputs": {"rng": "numpy.random.default_rng(1234).random(float32), levels (13,13,255) (26,26,255) (52,52,255) "
                             "drawn in that order from ONE generator", "class_num": 80, "nms_threshold": 0.5, "sigma": 0.5},
           "cases": []}
    for thr, reps in ((0.9, 5), (0.5, 1)):
        t_dec, dec, n = med(lambda: tools.decode(*lv, class_num=80, threshold=thr, version=3), reps)
        case = {"conf_threshold": thr, "candidates": int(dec.shape[0]), "runs": n, "decode_ms": round(t_dec, 3)}
        for name, fn in (("nms", lambda: tools.nms(dec, class_num=80, nms_threshold=0.5)),
                         ("diou_nms", lambda: tools.nms(dec, class_num=80, nms_threshold=0.5, iou_mode=2)),
                         ("soft_nms", lambda: tools.soft_nms(dec, class_num=80, nms_threshold=0.5, conf_threshold=thr,
                                                             sigma=0.5))):
            t, out, _ = med(fn, reps)
            case[name + "_ms"] = round(t, 3)
            case[name + "_kept"] = int(out.shape[0])
        res["cases"].append(case)
        print(case, flush=True)
    json.dump(res, open(os.path.join(HERE, "tools_timing.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
