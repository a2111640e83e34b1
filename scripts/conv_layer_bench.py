"""Per-shape conv kernel benchmark over the distinct conv shapes of a model at a given batch.
usage: python scripts/conv_layer_bench.py [v3|v4|v2|v1] [batch] [iters]"""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tf2_yolo_amd import graphs, ops

ver = sys.argv[1] if len(sys.argv) > 1 else "v3"
N = int(sys.argv[2]) if len(sys.argv) > 2 else 32
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 5
b = {"v3": lambda: graphs.build_yolov3((416, 416, 3), 80), "v4": lambda: graphs.build_yolov4((608, 608, 3), 80),
     "v2": lambda: graphs.build_yolov2((416, 416, 3), 20, [[1, 1]] * 5),
     "v1": lambda: graphs.build_yolov1_5((224, 224, 3), 1, 2)}[ver]()
shapes = collections.OrderedDict()
for u in b.units:
    if u.kind == "conv":
        key = (u.src.h, u.src.w, u.src.c, u.cout, u.k, u.stride, u.padding)
    elif u.kind == "head":
        key = (u.src.h, u.src.w, u.src.c, u.out.c, 1, 1, "same")
    else:
        continue
    shapes[key] = shapes.get(key, 0) + 1

def timeit(fn):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

tot = {"fwd": 0.0, "dgrad": 0.0, "wgrad": 0.0}
totf = 0.0
print(f"{'H':>4} {'Cin':>5} {'Cout':>5} k s  cnt |   M        K   | fwd ms  TF/s | dgrad ms TF/s | wgrad ms TF/s")
for (h, w, cin, cout, k, s, pad), cnt in shapes.items():
    d = ops.conv_desc((N, h, w, cin), cout, k, k, s, pad)
    x = torch.randn(N, h, w, cin, device="cuda")
    wt = torch.randn(cout, k, k, cin, device="cuda") * 0.05
    y = torch.empty(N, d.Ho, d.Wo, cout, device="cuda")
    dy = torch.randn(N, d.Ho, d.Wo, cout, device="cuda")
    dw = torch.zeros_like(wt)
    fl = 2.0 * N * d.Ho * d.Wo * cout * k * k * cin
    M0 = N * d.Ho * d.Wo
    if ops.planes_fwd_ok(cin, cout):
        xp = ops.split_planes(x, N * h * w, cin); wp = ops.split_planes(wt, cout, k * k * cin)
        t_f = timeit(lambda: ops.conv2d_fwd_planes(d, xp, wp, None, out=y))
    else:
        t_f = timeit(lambda: ops.conv2d_fwd(d, x, wt, None, out=y))
    if ops.planes_fwd_ok(cin, cout) and ops.planes_wgrad_ok(cin, cout, k * k, s):
        dyp = ops.split_planes(dy, M0, cout)
        t_w = timeit(lambda: ops.conv2d_wgrad_planes(d, xp, dyp, dw))
    else:
        t_w = timeit(lambda: ops.conv2d_wgrad(d, x, dy, dw))
    if cin % 32 == 0:
        wT = ops.filter_transpose(wt, cout, k * k, cin)
        dx = torch.empty_like(x)
        if ops.planes_dgrad_ok(cin, cout):
            dyp = ops.split_planes(dy, M0, cout); wTp = ops.split_planes(wT, cin, k * k * cout)
            t_d = timeit(lambda: ops.conv2d_dgrad_planes(d, dyp, wTp, dx=dx))
        else:
            t_d = timeit(lambda: ops.conv2d_dgrad(d, dy, wT, dx=dx))
    else:
        t_d = 0.0
    tot["fwd"] += t_f * cnt; tot["dgrad"] += t_d * cnt; tot["wgrad"] += t_w * cnt; totf += fl * cnt
    M = N * d.Ho * d.Wo
    print(f"{h:4d} {cin:5d} {cout:5d} {k} {s} {cnt:4d} | {M:8d} {k*k*cin:6d} | {t_f:6.3f} {fl/t_f/1e9:6.1f} | "
          f"{t_d:6.3f} {(fl/t_d/1e9 if t_d else 0):6.1f} | {t_w:6.3f} {fl/t_w/1e9:6.1f}")
print("totals ms:", {k: round(v, 2) for k, v in tot.items()}, "sum", round(sum(tot.values()), 2),
      "| fwd TF/s", round(totf / tot["fwd"] / 1e9, 1), "| all TF/s", round(3 * totf / sum(tot.values()) / 1e9, 1))
