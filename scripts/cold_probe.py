"""HBM-bound kernels with HOT operands (the same buffers again and again: they live in the 256 MB Infinity Cache) against COLD
ones (a ring of buffer sets several times the cache: every launch reads what it has not seen for a while) -- inside the
training step the operands of a layer were written a kernel ago by another kernel and are read once.
usage: cold_probe.py [layer: H,Cin,Cout,k]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf2_yolo_amd import ops
N = 32
RING = float(os.environ.get('COLD_RING_BYTES', '1.5e9'))   # bytes of buffer sets cycled through
layers = [(52, 256, 128, 1), (26, 512, 256, 1), (104, 128, 64, 1), (52, 128, 256, 3)]
if len(sys.argv) > 1:
    layers = [tuple(int(v) for v in sys.argv[1].split(","))]
def timed(fns, reps):
    for f in fns: f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for r in range(reps):
        for f in fns: f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (reps * len(fns)) * 1e3
for (H, cin, cout, k) in layers:
    d = ops.conv_desc((N, H, H, cin), cout, k, k, 1, "same")
    rows = N * H * H
    g = torch.Generator(device="cuda").manual_seed(1)
    w = torch.randn(cout, k * k * cin, device="cuda", generator=g) * 0.05
    wp = ops.split_planes(w, cout, k * k * cin)
    nbytes = rows * cin * 4 + rows * cout * 4
    nsets = max(2, int(RING // nbytes))
    sets = []
    for i in range(nsets):
        x = torch.randn(rows, cin, device="cuda", generator=g)
        sets.append((ops.split_planes(x, rows, cin), torch.empty((N, H, H, cout), device="cuda")))
        del x
    stats = torch.zeros(ops.BN_STAT_SLOTS * 2 * cout, device="cuda", dtype=torch.float64)
    amax = torch.zeros(cout, device="cuda", dtype=torch.int32)
    mk = lambda s: (lambda: ops.conv2d_fwd_planes(d, s[0], wp, None, out=s[1], stats=stats, absmax=amax))
    hot = timed([mk(sets[0])], 24)
    cold = timed([mk(s) for s in sets], max(2, 24 // nsets))
    print(f"conv fwd {H}x{H} {cin}->{cout} k{k} bs{N}: operands {nbytes / 1e6:.0f} MB, hot {hot:.1f} us ({nbytes / hot / 1e6:.2f} TB/s)  "
          f"cold ({nsets} sets) {cold:.1f} us ({nbytes / cold / 1e6:.2f} TB/s)", flush=True)
    # the BatchNorm forward pass that produces such planes: reads y (fp32), writes planes
    C = cout
    ys = [torch.randn(N, H, H, C, device="cuda", generator=g) for _ in range(max(2, int(RING // (rows * C * 8))))]
    pls = [torch.empty(ops.planes_bytes(rows, C), device="cuda", dtype=torch.uint8) for _ in ys]
    scale, shift = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    bnd = torch.zeros(4, device="cuda", dtype=torch.int32); bnd[0] = 0x42000000
    ob = torch.zeros(1, device="cuda")
    mkb = lambda i: (lambda: ops.bn_act_fwd(ys[i], C, scale, shift, 1, None, out=None, planes=pls[i], want_out=False, bn_bound=bnd[0:1],
                                            out_bound=ob))
    try:
        hotb = timed([mkb(0)], 24)
        coldb = timed([mkb(i) for i in range(len(ys))], max(2, 24 // len(ys)))
        nb = rows * C * 8
        print(f"   bn_act_fwd (y -> planes) {nb / 1e6:.0f} MB: hot {hotb:.1f} us ({nb / hotb / 1e6:.2f} TB/s)  cold {coldb:.1f} us ({nb / coldb / 1e6:.2f} TB/s)", flush=True)
    except Exception as e:
        print("   bn probe skipped:", repr(e)[:200])
    del sets, ys, pls
    torch.cuda.empty_cache()
