"""Run one conv shape repeatedly (for rocprofv3 counter collection).
usage: one_conv.py H Cin Cout k stride padding [batch] [iters] [mode fwd|dgrad|wgrad]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tf2_yolo_amd import ops
h, cin, cout, k, s = map(int, sys.argv[1:6])
pad = sys.argv[6]
N = int(sys.argv[7]) if len(sys.argv) > 7 else 32
iters = int(sys.argv[8]) if len(sys.argv) > 8 else 10
mode = sys.argv[9] if len(sys.argv) > 9 else "fwd"
d = ops.conv_desc((N, h, h, cin), cout, k, k, s, pad)
x = torch.randn(N, h, h, cin, device="cuda")
w = torch.randn(cout, k, k, cin, device="cuda") * 0.05
y = torch.empty(N, d.Ho, d.Wo, cout, device="cuda")
dy = torch.randn(N, d.Ho, d.Wo, cout, device="cuda")
if os.environ.get("YOLO_ONE_CONV_ZEROS") == "1":   # DVFS probe: same cycles, (almost) no switching energy
    x.zero_(); w.zero_(); dy.zero_()
dw = torch.zeros_like(w)
wT = ops.filter_transpose(w, cout, k * k, cin)
dx = torch.empty_like(x)
if mode.endswith("p"):   # planes kernels (pre-split operands, LDS-DMA)
    xp = ops.split_planes(x, N * h * h, cin); wp = ops.split_planes(w, cout, k * k * cin)
    dyp = ops.split_planes(dy, N * d.Ho * d.Wo, cout); wTp = ops.split_planes(wT, cin, k * k * cout)
else:
    xp = wp = dyp = wTp = None
fn = {"fwdp": lambda: ops.conv2d_fwd_planes(d, xp, wp, None, out=y),
      "dgradp": lambda: ops.conv2d_dgrad_planes(d, dyp, wTp, dx=dx),
      "wgradp": lambda: ops.conv2d_wgrad_planes(d, xp, dyp, dw),
      "fwd": lambda: ops.conv2d_fwd(d, x, w, None, out=y), "dgrad": lambda: ops.conv2d_dgrad(d, dy, wT, dx=dx),
      "wgrad": lambda: ops.conv2d_wgrad(d, x, dy, dw)}[mode]
fn(); torch.cuda.synchronize()
s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s0.record()
for _ in range(iters):
    fn()
e0.record(); torch.cuda.synchronize()
ms = s0.elapsed_time(e0) / iters
print(f"{mode} {ms:.4f} ms  {2.0*N*d.Ho*d.Wo*cout*k*k*cin/ms/1e9:.1f} TF/s")
