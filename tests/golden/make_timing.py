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


def model_output_case(tools):
    """BASELINE.json config 5 as README.md:296-334 runs it: the YOLOv3-416 network's OWN bs-1 prediction (random He-normal
    weights: tf2_yolo_amd.labels.synthetic_keras_weights(graph, 1234, residual_gamma=0.1), image rng(1234).random((1,416,416,3))) -> the
    reference's decode + the three NMS modes at conf_threshold .5. The prediction comes from the oracle's float32 CPU forward
    (oracle/models.py) of that network; bench.py builds the same network on the GPU (set_weights of the same arrays) and
    quotes these timings beside its own. The two predictions agree to ~1e-5, so the candidate counts may differ by a few
    rows that sit on the threshold."""
    import torch
    ROOT = os.path.dirname(os.path.dirname(HERE))
    sys.path.insert(0, ROOT)
    from oracle import models as OM
    from tf2_yolo_amd import graphs, labels
    w = labels.synthetic_keras_weights(graphs.build_yolov3((416, 416, 3), 80), 1234, residual_gamma=0.1)
    x = np.random.default_rng(1234).random((1, 416, 416, 3), dtype=np.float32)
    t0 = time.perf_counter()
    with torch.no_grad():
        outs, _ = OM.yolov3_forward({k: torch.from_numpy(v) for k, v in w.items()}, torch.from_numpy(x),
                                    graphs.V3_DEFAULT_ANCHORS, training=False)
    t_fwd = (time.perf_counter() - t0) * 1e3
    lv = [outs[2][0].numpy(), outs[1][0].numpy(), outs[0][0].numpy()]       # README.md:320-325 passes fine -> coarse
    t_dec, dec, n = med(lambda: tools.decode(*lv, class_num=80, threshold=0.5, version=3), 1)
    case = {"input": "YOLOv3-416 bs-1 prediction of the synthetic_keras_weights(seed 1234, residual_gamma 0.1) network on rng(1234) pixels, levels "
                     "passed fine -> coarse (52, 26, 13)", "conf_threshold": 0.5, "candidates": int(dec.shape[0]), "runs": n,
            "oracle_cpu_forward_ms_torch_fp32": round(t_fwd, 1), "decode_ms": round(t_dec, 3)}
    for name, fn in (("nms", lambda: tools.nms(dec, class_num=80, nms_threshold=0.5)),
                     ("diou_nms", lambda: tools.nms(dec, class_num=80, nms_threshold=0.5, iou_mode=2)),
                     ("soft_nms", lambda: tools.soft_nms(dec, class_num=80, nms_threshold=0.5, conf_threshold=0.5, sigma=0.5))):
        t, out, _ = med(fn, 1)
        case[name + "_ms"] = round(t, 3)
        case[name + "_kept"] = int(out.shape[0])
    return case


def main():
    tools = import_reference_tools()
    path = os.path.join(HERE, "tools_timing.json")
    if "--model-output-only" in sys.argv:   # add / refresh the C5 model-output case, keep the committed noise cases
        res = json.load(open(path))
        res["model_output"] = model_output_case(tools)
        print(res["model_output"], flush=True)
        json.dump(res, open(path, "w"), indent=1)
        return
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
    res["model_output"] = model_output_case(tools)
    print(res["model_output"], flush=True)
    json.dump(res, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()
