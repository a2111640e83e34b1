"""Planes (pre-split + LDS-DMA) conv kernels vs the register-staged split kernels: equality and timing.
usage: python scripts/planes_check.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tf2_yolo_amd import ops

N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
iters = 5

def timeit(fn):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

shapes = [(52, 128, 256, 3, 1, "same"), (52, 256, 128, 1, 1, "same"), (26, 256, 512, 3, 1, "same"),
          (13, 512, 1024, 3, 1, "same"), (104, 64, 128, 3, 1, "same"), (104, 128, 256, 3, 2, "same"),
          (13, 1024, 512, 1, 1, "same"), (52, 384, 128, 1, 1, "same"), (19, 48, 80, 3, 1, "same")]
torch.manual_seed(0)
print(f"{'H':>4} {'Cin':>5} {'Cout':>5} k s | fwd: split ms  planes ms  TF/s  maxdiff | dgrad: split ms planes ms TF/s maxdiff | split_x ms GB/s")
for (h, cin, cout, k, s, pad) in shapes:
    n = N if h != 19 else 3
    d = ops.conv_desc((n, h, h, cin), cout, k, k, s, pad)
    x = torch.randn(n, h, h, cin, device="cuda")
    w = torch.randn(cout, k, k, cin, device="cuda") * 0.05
    dy = torch.randn(n, d.Ho, d.Wo, cout, device="cuda")
    fl = 2.0 * n * d.Ho * d.Wo * cout * k * k * cin
    y0 = ops.conv2d_fwd(d, x, w)
    xp = ops.split_planes(x, n * h * h, cin)
    wp = ops.split_planes(w, cout, k * k * cin)
    y1 = ops.conv2d_fwd_planes(d, xp, wp)
    df = (y0 - y1).abs().max().item()
    t0 = timeit(lambda: ops.conv2d_fwd(d, x, w, out=y0))
    t1 = timeit(lambda: ops.conv2d_fwd_planes(d, xp, wp, out=y1))
    ts = timeit(lambda: ops.split_planes(x, n * h * h, cin, out=xp))
    wT = ops.filter_transpose(w, cout, k * k, cin)
    dx0 = ops.conv2d_dgrad(d, dy, wT)
    dyp = ops.split_planes(dy, n * d.Ho * d.Wo, cout)
    wTp = ops.split_planes(wT, cin, k * k * cout)
    dx1 = ops.conv2d_dgrad_planes(d, dyp, wTp)
    dd = (dx0 - dx1).abs().max().item()
    t2 = timeit(lambda: ops.conv2d_dgrad(d, dy, wT, dx=dx0))
    t3 = timeit(lambda: ops.conv2d_dgrad_planes(d, dyp, wTp, dx=dx1))
    print(f"{h:4d} {cin:5d} {cout:5d} {k} {s} | {t0:7.3f} {t1:7.3f} {fl/t1/1e9:7.1f} {df:9.2e} | {t2:7.3f} {t3:7.3f} {fl/t3/1e9:7.1f} {dd:9.2e} | "
          f"{ts:6.3f} {x.numel()*10/ts/1e6:6.0f}", flush=True)
