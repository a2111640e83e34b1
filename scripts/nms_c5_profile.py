"""C5 (BASELINE.json configs[4]): YOLOv3-416 bs-1 prediction with random weights, decode at 0.5, the three NMS modes --
candidates per class and wall time per call (run under rocprofv3 --kernel-trace --stats for the per-kernel times)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import yolov3
from tf2_yolo_amd import tools
y = yolov3.Yolo((416, 416, 3), [f"c{i}" for i in range(80)])
y.create_model(pretrained_body=None)
m = y.model
x = torch.from_numpy(np.random.default_rng(1234).random((1, 416, 416, 3), dtype=np.float32)).cuda()
outs = m.net.forward(x, training=False)
lv = [outs[2][0], outs[1][0], outs[0][0]]
dec = tools.decode_device(*lv, class_num=80, threshold=0.5, version=3)
cls = dec[:, 5].long()
cnt = torch.bincount(cls, minlength=80).cpu().numpy()
print("candidates", dec.shape[0], "classes with rows", int((cnt > 0).sum()), "largest classes", sorted(cnt.tolist(), reverse=True)[:12])
for name, fn in (("nms", lambda: tools.nms(dec, class_num=80, nms_threshold=0.5)),
                 ("diou", lambda: tools.nms(dec, class_num=80, nms_threshold=0.5, iou_mode=2)),
                 ("soft", lambda: tools.soft_nms(dec, class_num=80, nms_threshold=0.5, conf_threshold=0.5, sigma=0.5))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = fn()
    torch.cuda.synchronize()
    kc = torch.bincount(out[:, 5].long(), minlength=80).cpu().numpy()
    print(name, round((time.perf_counter() - t0) / 10 * 1e3, 3), "ms; kept", out.shape[0], "largest kept classes", sorted(kc.tolist(), reverse=True)[:6])
