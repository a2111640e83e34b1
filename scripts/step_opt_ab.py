"""Same process, same box, ONE model: blocks of K training steps alternating between settings of a library option
(yolo_set_option, read at launch time, so the replayed tape follows it), wall time per step of each block.
usage: step_opt_ab.py [--config c3|c4] [--k K] [--rounds R] --opt KEY  V0 V1 ...
e.g. the A/B bits of OPT_EXP (key 8): 0 = defaults, 8 = the loss kernel's chunk-ahead loader, 16 = forward launches with
BatchNorm statistics unsplit. (Only options whose results stay correct: a knock-out that leaves NaN in the network measures a
different power state, not the step.)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c3")
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--opt", type=int, default=8)
ap.add_argument("values", nargs="+", type=int)
a = ap.parse_args()
from tf2_yolo_amd import labels, optimizers, ops, graphs
ops.create_side_streams()
if a.config == "c4":
    import yolov4
    y = yolov4.Yolo((608, 608, 3), [f"c{i}" for i in range(80)])
    y.create_model(anchors=graphs.V4_DEFAULT_ANCHORS, pretrained_body=None)
    bs, hw = 16, 608
else:
    import yolov3
    y = yolov3.Yolo((416, 416, 3), [f"c{i}" for i in range(80)])
    y.create_model(pretrained_body=None, seed=1234)
    bs, hw = 32, 416
m = y.model
m.compile(optimizer=optimizers.Adam(learning_rate=1e-4), loss=y.loss())
x_h, ys_h = labels.synthetic_batch(np.random.default_rng(0), bs, (hw, hw), 80)
x = torch.from_numpy(x_h).cuda(); ys = [torch.from_numpy(t).cuda() for t in ys_h]
for _ in range(4):
    m.train_step_device(x, ys)
torch.cuda.synchronize()
for r in range(a.rounds):
    out = []
    for v in a.values:
        ops.set_option(a.opt, v)
        for _ in range(2):
            m.train_step_device(x, ys)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.k):
            m.train_step_device(x, ys)
        torch.cuda.synchronize()
        out.append(f"opt{a.opt}={v}: {(time.perf_counter() - t0) / a.k * 1e3:.2f}")
    ops.reset_options()
    print(f"round {r}: " + "   ".join(out) + "  ms/step", flush=True)
