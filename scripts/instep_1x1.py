"""The 1x1 layers of the C3 step, same box, same timing method (HIP events around each launch): inside the one-stream eager
step / alone on hot buffers / alone behind a 1 GB fill (cold caches). usage: instep_1x1.py"""
import os, sys
os.environ.setdefault("YOLO_STEP_MODE", "eager")
os.environ.setdefault("YOLO_BWD_OVERLAP", "0")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import yolov3
from tf2_yolo_amd import labels, optimizers, ops

yolo = yolov3.Yolo((416, 416, 3), [f"c{i}" for i in range(80)])
yolo.create_model(pretrained_body=None, seed=1234)
m = yolo.model
m.compile(optimizer=optimizers.Adam(learning_rate=1e-4), loss=yolo.loss())
x_h, ys_h = labels.synthetic_batch(np.random.default_rng(0), 32, (416, 416), 80)
x = torch.from_numpy(x_h).cuda(); ys = [torch.from_numpy(y).cuda() for y in ys_h]
for _ in range(3):
    m.train_step_device(x, ys)
torch.cuda.synchronize()
ops.TIMER = ops.KernelTimer()
STEPS = 4
for _ in range(STEPS):
    m.train_step_device(x, ys)
agg = ops.TIMER.summary(by_layer=True)
ops.TIMER = None
rows = []
for (name, layer), a in agg.items():
    if layer is None:
        continue
    which, H, W, cin, cout, k, s, N = layer
    rows.append((which, H, cin, cout, k, s, a["launches"] // STEPS, a["ms"] / a["launches"] * 1e3, name))
rows.sort(key=lambda r: -r[6] * r[7])

g = torch.Generator(device="cuda").manual_seed(1)
big = torch.empty(1 << 30, device="cuda", dtype=torch.uint8)


def alone(which, H, cin, cout, k, s):
    if which != "fwd" or s != 1:
        return None, None
    N = 32
    d = ops.conv_desc((N, H, H, cin), cout, k, k, 1, "same")
    r = N * H * H
    xp = ops.split_planes(torch.randn(r, cin, device="cuda", generator=g), r, cin)
    wp = ops.split_planes(torch.randn(cout, k * k * cin, device="cuda", generator=g) * 0.05, cout, k * k * cin)
    y = torch.empty((N, H, H, cout), device="cuda")
    st = torch.zeros(ops.BN_STAT_SLOTS * 2 * cout, device="cuda", dtype=torch.float64)
    am = torch.zeros(cout, device="cuda", dtype=torch.int32)
    f = lambda: ops.conv2d_fwd_planes(d, xp, wp, None, out=y, stats=st, absmax=am)
    def timed(pre, reps=12):
        f(); torch.cuda.synchronize()
        ev = []
        for _ in range(reps):
            if pre is not None: pre()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record(); ev.append((e0, e1))
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
        return t[len(t) // 2]
    return timed(None), timed(lambda: big.zero_())


print(f"{'pass':6s} {'H':>4s} {'Cin':>5s} {'Cout':>5s} k s  n/step  in-step us   alone hot   alone cold   kernel")
for r in rows[:40]:
    hot, cold = alone(*r[:6])
    print(f"{r[0]:6s} {r[1]:4d} {r[2]:5d} {r[3]:5d} {r[4]} {r[5]} {r[6]:6d} {r[7]:11.1f} "
          f"{(f'{hot:9.1f}' if hot else '        -')} {(f'{cold:11.1f}' if cold else '          -')}   {r[8]}", flush=True)
