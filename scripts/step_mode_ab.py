"""Same process, same model: blocks of K training steps alternating between the eager launches and the replayed step
(YOLO_STEP_MODE: tape or graph), wall time per step of each block. usage: step_mode_ab.py [K] [rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import yolov3
from tf2_yolo_amd import labels, optimizers
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
R = int(sys.argv[2]) if len(sys.argv) > 2 else 4
yolo = yolov3.Yolo((416, 416, 3), [f"c{i}" for i in range(80)])
yolo.create_model(pretrained_body=None, seed=1234)
m = yolo.model
m.compile(optimizer=optimizers.Adam(learning_rate=1e-4), loss=yolo.loss())
x_h, ys_h = labels.synthetic_batch(np.random.default_rng(0), 32, (416, 416), 80)
x = torch.from_numpy(x_h).cuda(); ys = [torch.from_numpy(y).cuda() for y in ys_h]
for _ in range(4):
    m.train_step_device(x, ys)
torch.cuda.synchronize()
print("recorded:", type(m._step_graphs).__name__ if m._step_graphs is not None else None, "mode", os.environ.get("YOLO_STEP_MODE", "tape"))
def block(eager):
    m._graphs_failed = eager
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        m.train_step_device(x, ys)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / K * 1e3
for r in range(R):
    a = block(False); b = block(True)
    print(f"round {r}: replay {a:.2f} ms/step   eager {b:.2f} ms/step", flush=True)
