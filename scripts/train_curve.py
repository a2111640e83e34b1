"""Loss curve of the first steps of YOLOv3-416 training on one fixed synthetic batch (Adam 1e-4), to compare the
arithmetic paths (YOLO_CONV_PLANES=1/0, YOLO_CONV_MODE=fp32). usage: python scripts/train_curve.py [batch] [steps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import yolov3
from tf2_yolo_amd import labels, optimizers

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 12
yolo = yolov3.Yolo((416, 416, 3), [f"c{i}" for i in range(80)])
yolo.create_model(pretrained_body=None, seed=1234)
m = yolo.model
m.compile(optimizer=optimizers.Adam(learning_rate=1e-4), loss=yolo.loss())
rng = np.random.default_rng(7)
x_h, ys_h = labels.synthetic_batch(rng, N, (416, 416), 80)
x = torch.from_numpy(x_h).cuda()
ys = [torch.from_numpy(v).cuda() for v in ys_h]
curve = []
for _ in range(steps):
    bufs, _ = m.train_step_device(x, ys)
    curve.append(round(float(sum(b[0].item() for b in bufs)), 3))
print(json.dumps(curve))
