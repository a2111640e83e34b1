"""cost of the per-channel max|y| atomics in the conv epilogue: the same launch with and without the absmax vector"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf2_yolo_amd import ops
N = 32
g = torch.Generator(device="cuda").manual_seed(1)
for (h, cin, cout, k) in [(52, 128, 256, 3), (26, 256, 512, 3), (13, 512, 1024, 3), (52, 256, 128, 1), (26, 512, 256, 1)]:
    d = ops.conv_desc((N, h, h, cin), cout, k, k, 1, "same")
    rows = N * h * h
    xp = ops.split_planes(torch.randn(rows, cin, device="cuda", generator=g), rows, cin)
    wp = ops.split_planes(torch.randn(cout, k * k * cin, device="cuda", generator=g) * 0.05, cout, k * k * cin)
    y = torch.empty((N, h, h, cout), device="cuda")
    st = torch.zeros(ops.BN_STAT_SLOTS * 2 * cout, device="cuda", dtype=torch.float64)
    am = torch.zeros(cout, device="cuda", dtype=torch.int32)
    res = {}
    def plain():
        ops.conv2d_fwd_planes(d, xp, wp, None, out=y)
    for _ in range(3): plain()
    torch.cuda.synchronize()
    ev = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plain(); e1.record(); ev.append((e0, e1))
    torch.cuda.synchronize()
    t = sorted(p_.elapsed_time(q_) * 1e3 for p_, q_ in ev)
    res["no statistics at all"] = t[len(t) // 2]
    for name, a in (("with absmax", am), ("without", None), ("with absmax, zeroed each launch", "z")):
        def f():
            if a is "z":
                am.zero_()
                ops.conv2d_fwd_planes(d, xp, wp, None, out=y, stats=st, absmax=am)
            else:
                ops.conv2d_fwd_planes(d, xp, wp, None, out=y, stats=st, absmax=a)
        for _ in range(3): f()
        torch.cuda.synchronize()
        ev = []
        for _ in range(20):
            if a is "z": am.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); ops.conv2d_fwd_planes(d, xp, wp, None, out=y, stats=st, absmax=(am if a is not None else None)); e1.record()
            ev.append((e0, e1))
        torch.cuda.synchronize()
        t = sorted(p.elapsed_time(q) * 1e3 for p, q in ev)
        res[name] = t[len(t) // 2]
    print(f"{h}x{h} {cin}->{cout} k{k}: " + ", ".join(f"{n} {v:.1f} us" for n, v in res.items()), flush=True)
