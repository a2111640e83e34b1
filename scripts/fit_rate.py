"""PCIe-inclusive training rate: `Model.fit` on HOST (NumPy) arrays at the bench configuration
(YOLOv3 416x416, bs 32, C=80), streamed (feeder.HostFeeder) vs the plain batch-by-batch loop.
DESIGN.md quotes this beside bench.py's device-resident `value`.  usage: fit_rate.py [images] [epochs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import yolov3
from tf2_yolo_amd import labels, optimizers

n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
epochs = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.default_rng(0)
xb, ysb = labels.synthetic_batch(rng, 32, (416, 416), 80)
reps = n // 32
x = np.concatenate([xb] * reps)
ys = [np.concatenate([y] * reps) for y in ysb]
print(f"host dataset: {n} images, x {x.nbytes/2**20:.0f} MiB, labels {sum(y.nbytes for y in ys)/2**20:.0f} MiB", flush=True)
for mode in ("1", "0"):
    os.environ["YOLO_FIT_PIPELINE"] = mode
    yolo = yolov3.Yolo((416, 416, 3), [f"c{i}" for i in range(80)])
    yolo.create_model(pretrained_body=None, seed=1234)
    yolo.model.compile(optimizer=optimizers.Adam(learning_rate=1e-4), loss=yolo.loss())
    yolo.model.fit(x[:64], [y[:64] for y in ys], batch_size=32, epochs=1, verbose=0)   # warm-up / allocation
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    h = yolo.model.fit(x, ys, batch_size=32, epochs=epochs, verbose=0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"YOLO_FIT_PIPELINE={mode}: {n * epochs / dt:.1f} images/s from host arrays "
          f"({dt / (reps * epochs) * 1e3:.1f} ms/batch), loss {h.history['loss'][-1]:.3f}", flush=True)
