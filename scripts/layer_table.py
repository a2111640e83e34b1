"""Per-layer account of the planes conv launches of the YOLOv3-416 bs-32 training step: device time INSIDE the two-stream
step (HIP events around each launch, ops.KernelTimer) beside the same launch run ALONE (back to back on an idle chip) --
the table VERDICT r02 #6 asks for (in-step vs standalone rate of the window / patch / filter-gradient kernels).
usage: python scripts/layer_table.py out.json [c3|c4]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import yolov3
from tf2_yolo_amd import labels, ops, optimizers

out_path = sys.argv[1] if len(sys.argv) > 1 else None
CONFIG = sys.argv[2] if len(sys.argv) > 2 else "c3"      # c3: YOLOv3-416 bs 32 (default), c4: YOLOv4-608 bs 16
if CONFIG == "c4":
    import yolov4
    from tf2_yolo_amd import graphs
    N, HW, C = 16, 608, 80
    yolo = yolov4.Yolo((HW, HW, 3), [f"c{i}" for i in range(C)])
    yolo.create_model(anchors=graphs.V4_DEFAULT_ANCHORS, pretrained_body=None)
else:
    N, HW, C = 32, 416, 80
    yolo = yolov3.Yolo((HW, HW, 3), [f"c{i}" for i in range(C)])
    yolo.create_model(pretrained_body=None, seed=1234)
m = yolo.model
m.compile(optimizer=optimizers.Adam(learning_rate=1e-4), loss=yolo.loss())
rng = np.random.default_rng(1234)
x_h, ys_h = (labels.synthetic_batch(rng, N, (HW, HW), C, levels=3, finest_stride=8) if CONFIG == "c4"
            else labels.synthetic_batch(rng, N, (HW, HW), C))
x = torch.from_numpy(x_h).cuda()
ys = [torch.from_numpy(y).cuda() for y in ys_h]
for _ in range(3):
    m.train_step_device(x, ys)
torch.cuda.synchronize()
STEPS = 4
t = ops.KernelTimer()
ops.TIMER = t
for _ in range(STEPS):
    m.train_step_device(x, ys)
torch.cuda.synchronize()
ops.TIMER = None
instep = t.summary(by_layer=True)
# the same with the filter-gradient stream off (every kernel has the chip to itself but runs inside the step's power state)
m.net._overlap_wgrad = False
t = ops.KernelTimer()
ops.TIMER = t
for _ in range(2):
    m.train_step_device(x, ys)
torch.cuda.synchronize()
ops.TIMER = None
serial = t.summary(by_layer=True)
m.net._overlap_wgrad = True
del m, yolo
torch.cuda.empty_cache()


def standalone(layer, reps=12):
    which, H, W, cin, cout, k, s, n = layer
    pad = "same" if s == 1 else "darknet_s2"
    d = ops.conv_desc((n, H, W, cin), cout, k, k, s, pad)
    g = torch.Generator(device="cuda").manual_seed(7)
    xp = ops.split_planes(torch.randn(n * H * W, cin, device="cuda", generator=g), n * H * W, cin)
    rows_o = n * d.Ho * d.Wo
    dyp = ops.split_planes(torch.randn(rows_o, cout, device="cuda", generator=g), rows_o, cout)
    w = torch.randn(cout, k * k * cin, device="cuda", generator=g) * 0.05
    if which == "fwd":
        wp = ops.split_planes(w, cout, k * k * cin)
        out = torch.empty((n, d.Ho, d.Wo, cout), device="cuda")
        stats = torch.zeros(ops.BN_STAT_SLOTS * 2 * cout, device="cuda", dtype=torch.float64)
        amax = torch.zeros(cout, device="cuda", dtype=torch.int32)
        fn = lambda: ops.conv2d_fwd_planes(d, xp, wp, None, out=out, stats=stats, absmax=amax)
    elif which == "dgrad":
        wT = ops.filter_transpose(w.reshape(-1), cout, k * k, cin)
        wTp = ops.split_planes(wT, cin, k * k * cout)
        dx = torch.empty((n, H, W, cin), device="cuda")
        fn = lambda: ops.conv2d_dgrad_planes(d, dyp, wTp, dx=dx)
    else:
        dw = torch.zeros(cout * k * k * cin, device="cuda")
        fn = lambda: ops.conv2d_wgrad_planes(d, xp, dyp, dw)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


rows = []
for (name, layer), a in sorted(instep.items(), key=lambda kv: -kv[1]["ms"]):
    if layer is None or layer[3] % 16 or layer[4] % 16:
        continue      # (the 255-channel heads run on padded copies: not reproducible from the descriptor alone)
    us_in = a["ms"] / a["launches"] * 1e3
    b = serial.get((name, layer))
    us_ser = b["ms"] / b["launches"] * 1e3 if b else None
    us_alone = standalone(layer)
    fl = a["flops"] / a["launches"]
    rows.append({"kernel": name, "pass": layer[0], "H": layer[1], "W": layer[2], "Cin": layer[3], "Cout": layer[4], "k": layer[5],
                 "stride": layer[6], "launches_per_step": a["launches"] // STEPS, "us_in_step": round(us_in, 1),
                 "us_step_one_stream": None if us_ser is None else round(us_ser, 1), "us_alone": round(us_alone, 1),
                 "ms_per_step": round(a["ms"] / STEPS, 3), "tflops_in_step": round(fl / us_in / 1e6, 1),
                 "tflops_alone": round(fl / us_alone / 1e6, 1)})
by_kernel = {}
for r in rows:
    k = by_kernel.setdefault(r["kernel"], {"ms_in_step": 0.0, "ms_one_stream": 0.0, "ms_alone": 0.0, "gflop": 0.0})
    k["ms_in_step"] += r["us_in_step"] * r["launches_per_step"] / 1e3
    k["ms_one_stream"] += (r["us_step_one_stream"] or 0) * r["launches_per_step"] / 1e3
    k["ms_alone"] += r["us_alone"] * r["launches_per_step"] / 1e3
    k["gflop"] += r["tflops_alone"] * r["us_alone"] * r["launches_per_step"] / 1e3
for k in by_kernel.values():
    for f in ("ms_in_step", "ms_one_stream", "ms_alone"):
        k["frac_of_833_" + f[3:]] = round(k["gflop"] / k[f] / 833.3, 3) if k[f] else None
        k[f] = round(k[f], 3)
res = {"what": ("YOLOv4-608 bs 16" if CONFIG == "c4" else "YOLOv3-416 bs 32") + " training step, planes conv launches: us per launch inside the two-stream step, inside the step with "
               "the filter-gradient stream off, and alone (12 back-to-back launches of the same shape on random data)",
       "device": torch.cuda.get_device_name(0), "by_kernel": by_kernel, "layers": rows}
print(json.dumps(res["by_kernel"], indent=1))
for r in rows[:40]:
    print(r)
if out_path:
    json.dump(res, open(out_path, "w"), indent=1)
