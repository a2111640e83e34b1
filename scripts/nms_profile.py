"""decode + the three NMS modes on BASELINE.md's 131 304-candidate input (for rocprofv3 --kernel-trace --stats)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tf2_yolo_amd import tools
rng = np.random.default_rng(1234)
lv = [torch.from_numpy(rng.random((g, g, 255), dtype=np.float32)).cuda() for g in (13, 26, 52)]
dec = tools.decode_device(*lv, class_num=80, threshold=0.5, version=3)
print("rows", dec.shape[0])
for name, fn in (("nms", lambda: tools.nms(dec, class_num=80, nms_threshold=0.5)),
                 ("diou", lambda: tools.nms(dec, class_num=80, nms_threshold=0.5, iou_mode=2)),
                 ("soft", lambda: tools.soft_nms(dec, class_num=80, nms_threshold=0.5, conf_threshold=0.5, sigma=0.5))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    print(name, round((time.perf_counter() - t0) / 3 * 1e3, 3), "ms", out.shape[0])
