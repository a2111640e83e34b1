"""Golden files for array_to_json / array_to_xml, written by EXECUTING THE REFERENCE's own utils/tools.py
(build container only). The file BYTES are stored as uint8 arrays in tests/golden/writers_golden.npz.
Run:  python -B tests/golden/make_writers_golden.py"""
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_inputs                                  # noqa: E402
from make_golden import import_reference_tools      # noqa: E402

CASES = [("w0", 0, dict(conf_threshold=0.4, nms_mode=1)), ("w1", 3, dict(conf_threshold=0.3, nms_mode=0)),
         ("w2", 5, dict(conf_threshold=0.3, nms_mode=2, nms_threshold=0.4, nms_sigma=0.6)),
         ("w3", 1, dict(conf_threshold=0.999, nms_mode=3))]   # w3: nothing detected
NAMES = ["a", "b", "c"]
IMG_SIZE = (320, 480)


def main():
    tools = import_reference_tools()
    _, lv0, lv1 = gen_inputs.measurement_inputs()
    out = {}
    d = tempfile.mkdtemp()
    for key, img, kw in CASES:
        pj, px = os.path.join(d, key + ".json"), os.path.join(d, key + ".xml")
        tools.array_to_json(pj, IMG_SIZE, lv0[img], lv1[img], class_names=NAMES, version=3, **kw)
        tools.array_to_xml(px, IMG_SIZE, lv0[img], lv1[img], class_names=NAMES, version=3, **kw)
        out[key + "_json"] = np.frombuffer(open(pj, "rb").read(), dtype=np.uint8)
        out[key + "_xml"] = np.frombuffer(open(px, "rb").read(), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "writers_golden.npz"), **out)
    print({k: v.size for k, v in out.items()})


if __name__ == "__main__":
    main()
