"""YOLOv3-416 bs-1 inference forward, N eager passes (for rocprofv3 --kernel-trace --stats). usage: infer_bs1.py [n]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import yolov3
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
y = yolov3.Yolo((416, 416, 3), [f"c{i}" for i in range(80)])
y.create_model(pretrained_body=None)
x = torch.from_numpy(np.random.default_rng(1234).random((1, 416, 416, 3), dtype=np.float32)).cuda()
net = y.model.net
for _ in range(3):
    net.forward(x, training=False)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(n):
    net.forward(x, training=False)
torch.cuda.synchronize()
print("eager ms", (time.perf_counter() - t0) / n * 1e3)
