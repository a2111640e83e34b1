"""Host-side cost of enqueuing one training step (time until train_step_device returns, device not awaited)
vs the device-bound step time. usage: host_enqueue.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import yolov3
from tf2_yolo_amd import labels, optimizers
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
yolo = yolov3.Yolo((416, 416, 3), [f"c{i}" for i in range(80)])
yolo.create_model(pretrained_body=None, seed=1234)
m = yolo.model
m.compile(optimizer=optimizers.Adam(learning_rate=1e-4), loss=yolo.loss())
x_h, ys_h = labels.synthetic_batch(np.random.default_rng(0), 32, (416, 416), 80)
x = torch.from_numpy(x_h).cuda(); ys = [torch.from_numpy(y).cuda() for y in ys_h]
for _ in range(3):
    m.train_step_device(x, ys)
torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(steps):
    a = time.perf_counter()
    m.train_step_device(x, ys)
    host.append(time.perf_counter() - a)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
solo = []
for _ in range(steps):      # the same with an idle device in front of every step: pure host cost of producing the launches
    torch.cuda.synchronize()
    a = time.perf_counter()
    m.train_step_device(x, ys)
    solo.append(time.perf_counter() - a)
torch.cuda.synchronize()
print(f"step launch mode: {type(getattr(m, '_step_graphs', None)).__name__ if getattr(m, '_step_graphs', None) is not None else 'eager'} "
      f"(YOLO_STEP_MODE={os.environ.get('YOLO_STEP_MODE', 'tape')}); host time per step with an idle device in front: "
      f"median {np.median(solo)*1e3:.2f} ms (min {min(solo)*1e3:.2f})")
print(f"host enqueue per step: median {np.median(host)*1e3:.1f} ms (min {min(host)*1e3:.1f}, max {max(host)*1e3:.1f}); "
      f"loop returned after {(t1-t0)*1e3:.0f} ms, device drained after {(t2-t0)*1e3:.0f} ms "
      f"({(t2-t0)/steps*1e3:.1f} ms/step)")
