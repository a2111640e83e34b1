"""Generate golden vectors by EXECUTING THE REFERENCE's own NumPy code (build container only).

Run:  python -B tests/golden/make_golden.py
Imports /root/reference/utils/tools.py after registering empty stub modules for its absent,
unused-by-these-functions imports (cv2, bs4, imgaug, tensorflow.keras.utils.Sequence), exactly
as SURVEY.md section 8c records. Only inputs and outputs (data) are written to tests/golden/;
no reference source or bytecode is copied. The reference never travels to the GPU box.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def import_reference_tools():
    import matplotlib
    matplotlib.use("Agg")
    _stub("cv2")
    _stub("bs4", BeautifulSoup=object)
    _stub("imgaug")
    _stub("imgaug.augmentables")
    _stub("imgaug.augmentables.bbs", BoundingBox=object, BoundingBoxesOnImage=object)
    _stub("tensorflow")
    _stub("tensorflow.keras")
    _stub("tensorflow.keras.utils", Sequence=object)
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    from utils import tools  # noqa: the reference's utils/tools.py
    return tools


def main():
    tools = import_reference_tools()
    sys.path.insert(0, HERE)
    import gen_inputs
    out = {}
    case = 0
    for key, C, thr, lv in gen_inputs.decode_cases():   # README passes levels fine -> coarse
        dec = tools.decode(*lv, class_num=C, threshold=thr, version=3)
        if dec.size == 0:
            dec = dec.reshape(0, 7)
        out[f"{key}_decode"] = dec
        if len(dec):
            scores = dec[:, 4] * dec[:, 6]
            for c in range(C):   # ties only matter inside a class (argsort instability)
                sc = scores[dec[:, 5] == c]
                assert len(np.unique(sc)) == len(sc), "score tie: pick another seed"
            out[f"{key}_nms"] = tools.nms(dec, class_num=C, nms_threshold=0.5)
            out[f"{key}_diou"] = tools.nms(dec, class_num=C, nms_threshold=0.5, iou_mode=2)
            out[f"{key}_soft"] = tools.soft_nms(dec, class_num=C, nms_threshold=0.5,
                                                conf_threshold=thr, sigma=0.5)
        case += 1
    misc = gen_inputs.misc_inputs()
    # YOLOv1-layout decode (version=1) and YOLOv2 (version=2)
    out["v1_decode"] = tools.decode(misc["v1_lv"], class_num=4, threshold=0.4, version=1)
    out["v2_decode"] = tools.decode(misc["v2_lv"], class_num=20, threshold=0.8, version=2)
    # float64 label tensors (vis_img on ground truth decodes float64 arrays)
    lab = misc["label52"]
    l26 = tools.down2xlabel(lab)
    l13 = tools.down2xlabel(l26)
    out["label26"] = l26
    out["label13"] = l13
    out["label52_decode"] = tools.decode(lab[0], class_num=3, threshold=0.5, version=3)
    out["binary_weight52"] = tools.get_class_weight(lab[..., 4:5], "binary")
    for m in ("alpha", "log", "effective"):
        out[f"class_weight_{m}"] = tools.get_class_weight(lab[..., 5:], m)
    # pairwise IoU / DIoU matrices
    boxes = misc["iou_boxes"]
    out["iou_mat"] = tools.cal_iou(boxes.reshape(-1, 1, 5), boxes.reshape(1, -1, 5), mode=1)
    out["diou_mat"] = tools.cal_iou(boxes.reshape(-1, 1, 5), boxes.reshape(1, -1, 5), mode=2)
    np.savez_compressed(os.path.join(HERE, "tools_golden.npz"), **out)
    print("cases:", case, "arrays:", len(out))


if __name__ == "__main__":
    main()
