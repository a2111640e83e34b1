import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from tf2_yolo_amd import labels
from tf2_yolo_amd.optimizers import Adam
import yolov2
lr = float(sys.argv[1]); steps = int(sys.argv[2])
rng = np.random.default_rng(2)
y = yolov2.Yolo((64, 64, 3), ["a", "b", "c"])
y.create_model()
x, ys = labels.synthetic_batch(rng, 4, (64, 64), 3, levels=1, finest_stride=32)
y.model.compile(optimizer=Adam(learning_rate=lr), loss=y.loss())
out = []
for i in range(steps):
    out.append(round(float(y.model.train_on_batch(x, ys[0])), 2))
print(lr, out, flush=True)
