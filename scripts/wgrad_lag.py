"""How long does the compute stream wait for the filter-gradient stream at the end of backward? (eager steps, C3 at bs 32)
Events: main stream at the end of backward BEFORE the join, the filter-gradient stream's end, backward start.
usage: python scripts/wgrad_lag.py"""
import os, sys
os.environ.setdefault("YOLO_STEP_MODE", "eager")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import yolov3
from tf2_yolo_amd import labels, optimizers

y = yolov3.Yolo((416, 416, 3), [f"c{i}" for i in range(80)])
y.create_model(pretrained_body=None, seed=1234)
m = y.model
m.compile(optimizer=optimizers.Adam(learning_rate=1e-4), loss=y.loss())
rng = np.random.default_rng(1234)
xh, ysh = labels.synthetic_batch(rng, 32, (416, 416), 80)
x = torch.from_numpy(xh).cuda()
ys = [torch.from_numpy(a).cuda() for a in ysh]
net = m.net
rec = {}
orig_join, orig_bwd = net._join_wgrad, net.backward


def join():
    rec["main_end"] = torch.cuda.Event(enable_timing=True)
    rec["main_end"].record(torch.cuda.current_stream())
    if net._wgrad_stream is not None:
        rec["side_end"] = torch.cuda.Event(enable_timing=True)
        rec["side_end"].record(net._wgrad_stream)
    orig_join()
    rec["joined"] = torch.cuda.Event(enable_timing=True)
    rec["joined"].record(torch.cuda.current_stream())


def bwd(d):
    rec["start"] = torch.cuda.Event(enable_timing=True)
    rec["start"].record(torch.cuda.current_stream())
    return orig_bwd(d)


net._join_wgrad, net.backward = join, bwd
for _ in range(5):
    m.train_step_device(x, ys)
torch.cuda.synchronize()
out = []
for _ in range(6):
    t0 = torch.cuda.Event(enable_timing=True); t0.record()
    m.train_step_device(x, ys)
    t1 = torch.cuda.Event(enable_timing=True); t1.record()
    torch.cuda.synchronize()
    out.append((t0.elapsed_time(t1), t0.elapsed_time(rec["start"]), rec["start"].elapsed_time(rec["main_end"]),
                rec["start"].elapsed_time(rec["side_end"]), rec["start"].elapsed_time(rec["joined"])))
for o in out:
    print("step %.2f ms | forward+loss %.2f | backward: compute stream done at %.2f, filter-gradient stream at %.2f, joined %.2f" % o)
