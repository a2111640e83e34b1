"""Why is an HBM-bound 1x1 layer 40-80 % slower inside the training step than alone (profiles/r03_layer_table.json)?
The 52x52 256->128 forward launch timed with HIP events (a) alone on hot buffers, (b) right behind an MFMA-heavy 3x3 launch
that touches other buffers, (c) behind the 3x3 launch AND the BatchNorm pass that writes its operand (the data flow of the
step), (d) behind a 1 GB fill (cold caches, idle matrix cores), (e) behind the fill and the 3x3 launch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf2_yolo_amd import ops

N, H = 32, 52
g = torch.Generator(device="cuda").manual_seed(1)
rows = N * H * H


def planes_of(r, c, scale=1.0):
    x = torch.randn(r, c, device="cuda", generator=g) * scale
    return ops.split_planes(x, r, c)


d3 = ops.conv_desc((N, H, H, 128), 256, 3, 3, 1, "same")
d1 = ops.conv_desc((N, H, H, 256), 128, 1, 1, 1, "same")
x3p, w3p = planes_of(rows, 128), planes_of(256, 1152, 0.05)
x1p, w1p = planes_of(rows, 256), planes_of(128, 256, 0.05)
y3 = torch.empty((N, H, H, 256), device="cuda")
y1 = torch.empty((N, H, H, 128), device="cuda")
st3 = torch.zeros(ops.BN_STAT_SLOTS * 2 * 256, device="cuda", dtype=torch.float64)
st1 = torch.zeros(ops.BN_STAT_SLOTS * 2 * 128, device="cuda", dtype=torch.float64)
am3 = torch.zeros(256, device="cuda", dtype=torch.int32)
am1 = torch.zeros(128, device="cuda", dtype=torch.int32)
scale, shift = torch.ones(256, device="cuda"), torch.zeros(256, device="cuda")
bnd = torch.zeros(4, device="cuda", dtype=torch.int32); bnd[0] = 0x42000000
ob = torch.zeros(1, device="cuda")
big = torch.empty(1 << 30, device="cuda", dtype=torch.uint8)

conv3 = lambda: ops.conv2d_fwd_planes(d3, x3p, w3p, None, out=y3, stats=st3, absmax=am3)
conv1 = lambda: ops.conv2d_fwd_planes(d1, x1p, w1p, None, out=y1, stats=st1, absmax=am1)
bn = lambda: ops.bn_act_fwd(y3, 256, scale, shift, 1, None, out=None, planes=x1p, want_out=False, bn_bound=bnd[0:1], out_bound=ob)
fill = lambda: big.zero_()


def run(name, seq, reps=20):
    for f in seq: f()
    torch.cuda.synchronize()
    evs = []
    for r in range(reps):
        row = []
        for f in seq:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); f(); e1.record()
            row.append((e0, e1))
        evs.append(row)
    torch.cuda.synchronize()
    out = []
    for i in range(len(seq)):
        ts = sorted(r[i][0].elapsed_time(r[i][1]) * 1e3 for r in evs)
        out.append(ts[len(ts) // 2])
    print(f"{name:60s} " + "  ".join(f"{t:7.1f}" for t in out) + "  us (median per launch, in sequence order)", flush=True)


run("(a) 1x1 alone, hot", [conv1])
run("(a') 3x3 alone", [conv3])
run("(b) 3x3 (other buffers), 1x1", [conv3, conv1])
run("(b') 3x3, 3x3, 3x3, 1x1", [conv3, conv3, conv3, conv1])
run("(c) 3x3, bn (writes the 1x1 operand), 1x1", [conv3, bn, conv1])
run("(c') bn, 1x1", [bn, conv1])
run("(d) 1 GB fill, 1x1", [fill, conv1])
run("(e) 1 GB fill, 3x3, 1x1", [fill, conv3, conv1])
run("(f) 1x1, 1x1, 1x1, 1x1", [conv1, conv1, conv1, conv1])
