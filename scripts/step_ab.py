"""Same process, same box: models built under different environment settings (read by Network.__init__ / Model), blocks of K
training steps alternating between them, wall time per step of each block.
usage: step_ab.py [--config c3|c4] [--k K] [--rounds R] NAME:VAR=VAL[,VAR=VAL...] NAME:... (an empty setting list = defaults)"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="c3")
ap.add_argument("--k", type=int, default=10)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("variants", nargs="+")
a = ap.parse_args()
from tf2_yolo_amd import labels, optimizers, ops
ops.create_side_streams()


def build():
    if a.config == "c4":
        import yolov4
        y = yolov4.Yolo((608, 608, 3), [f"c{i}" for i in range(80)])
        from tf2_yolo_amd import graphs
        y.create_model(anchors=graphs.V4_DEFAULT_ANCHORS, pretrained_body=None)
        bs, hw = 16, 608
    else:
        import yolov3
        y = yolov3.Yolo((416, 416, 3), [f"c{i}" for i in range(80)])
        y.create_model(pretrained_body=None, seed=1234)
        bs, hw = 32, 416
    m = y.model
    m.compile(optimizer=optimizers.Adam(learning_rate=1e-4), loss=y.loss())
    return m, bs, hw


models = []
for v in a.variants:
    name, _, sets = v.partition(":")
    kv = dict(s.split("=", 1) for s in sets.split(",") if s)
    saved = {k: os.environ.get(k) for k in kv}
    os.environ.update(kv)
    m, bs, hw = build()
    x_h, ys_h = labels.synthetic_batch(np.random.default_rng(0), bs, (hw, hw), 80)
    x = torch.from_numpy(x_h).cuda(); ys = [torch.from_numpy(t).cuda() for t in ys_h]
    for _ in range(4):
        bufs, _ = m.train_step_device(x, ys)
    torch.cuda.synchronize()
    print(f"{name}: {kv} loss after 4 steps {sum(float(b[0].item()) for b in bufs):.4f}", flush=True)
    for k, old in saved.items():
        if old is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = old
    models.append((name, m, x, ys))
for r in range(a.rounds):
    out = []
    for name, m, x, ys in models:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.k):
            m.train_step_device(x, ys)
        torch.cuda.synchronize()
        out.append(f"{name} {(time.perf_counter() - t0) / a.k * 1e3:.2f}")
    print(f"round {r}: " + "   ".join(out) + "  ms/step", flush=True)
