"""wgrad planes kernel vs the register-staged split wgrad kernel: agreement and timing per YOLOv3 layer shape."""
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
shapes = [(208, 32, 64, 3, 1), (104, 64, 128, 3, 1), (104, 128, 64, 1, 1), (52, 128, 256, 3, 1), (52, 256, 128, 1, 1),
          (26, 256, 512, 3, 1), (26, 512, 256, 1, 1), (13, 512, 1024, 3, 1), (13, 1024, 512, 1, 1), (104, 128, 256, 3, 2)]
print("   H   Cin  Cout k s | split ms  planes ms   TF/s   relerr")
for (h, cin, cout, k, s) in shapes:
    d = ops.conv_desc((N, h, h, cin), cout, k, k, s, "same")
    x = torch.randn(N, h, h, cin, device="cuda")
    dy = torch.randn(N, d.Ho, d.Wo, cout, device="cuda")
    fl = 2.0 * N * d.Ho * d.Wo * cout * k * k * cin
    dw0 = torch.zeros(cout, k, k, cin, device="cuda"); dw1 = torch.zeros_like(dw0)
    ops.conv2d_wgrad(d, x, dy, dw0)
    xp = ops.split_planes(x, N * h * h, cin); dyp = ops.split_planes(dy, N * d.Ho * d.Wo, cout)
    ops.conv2d_wgrad_planes(d, xp, dyp, dw1)
    err = ((dw0 - dw1).abs().max() / dw0.abs().max()).item()
    t0 = timeit(lambda: ops.conv2d_wgrad(d, x, dy, dw0)); t1 = timeit(lambda: ops.conv2d_wgrad_planes(d, xp, dyp, dw1))
    print(f"{h:4d} {cin:5d} {cout:5d} {k} {s} | {t0:7.3f} {t1:7.3f} {fl/t1/1e9:7.1f} {err:9.2e}", flush=True)
