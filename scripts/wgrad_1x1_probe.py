"""1x1 filter gradients alone, with the slabs workspace (the step's form): N launches of each layer for a rocprofv3 kernel trace.
usage: wgrad_1x1_probe.py [reps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf2_yolo_amd import ops
ops.ensure_wgrad_workspace()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = 32
for (h, cin, cout) in [(52, 256, 128), (26, 512, 256), (13, 1024, 512), (104, 128, 64)]:
    d = ops.conv_desc((N, h, h, cin), cout, 1, 1, 1, "same")
    g = torch.Generator(device="cuda").manual_seed(1)
    rows = N * h * h
    xp = ops.split_planes(torch.randn(rows, cin, device="cuda", generator=g), rows, cin)
    dyp = ops.split_planes(torch.randn(rows, cout, device="cuda", generator=g), rows, cout)
    dw = torch.zeros(cout * cin, device="cuda")
    big = torch.empty(1 << 29, device="cuda", dtype=torch.uint8)
    for mode in ("hot", "cold"):
        for _ in range(2):
            ops.conv2d_wgrad_planes(d, xp, dyp, dw)
        torch.cuda.synchronize()
        ev = []
        for _ in range(reps):
            if mode == "cold":
                big.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.conv2d_wgrad_planes(d, xp, dyp, dw); e1.record(); ev.append((e0, e1))
        torch.cuda.synchronize()
        t = sorted(a.elapsed_time(b) * 1e3 for a, b in ev)
        print(f"wgrad 1x1 {h}x{h} {cin}->{cout} {mode}: {t[len(t)//2]:.1f} us (partials + reduce, HIP events)", flush=True)
