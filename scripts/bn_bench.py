"""Bandwidth of the BN/activation kernels on YOLOv3 activation shapes. usage: python scripts/bn_bench.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tf2_yolo_amd import ops

def timeit(fn, iters=20):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

print("  H    C |  fwd ms GB/s | fwd+planes ms GB/s | reduce ms GB/s | apply ms GB/s | apply planes-only ms GB/s")
for h, C in [(208, 64), (104, 128), (52, 256), (26, 512), (13, 1024), (52, 128), (416, 32)]:
    N = 32
    P = N * h * h
    x = torch.randn(N, h, h, C, device="cuda"); dout = torch.randn_like(x) * 1e-3; out = torch.empty_like(x)
    stats = torch.zeros(64 * 2 * C, device="cuda", dtype=torch.float64); red = torch.zeros(513 * 2 * C, device="cuda", dtype=torch.float64)
    f = lambda: torch.empty(C, device="cuda")
    scale, shift, smean, sinv = f(), f(), f(), f()
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    aux = torch.zeros(72, device="cuda", dtype=torch.int32)
    ops.bn_stats(x, C, stats); ops.bn_finalize(stats, P, C, g, b, None, None, scale, shift, smean, sinv, bound=aux[0:1])
    pl = torch.zeros(ops.planes_bytes(P, C), device="cuda", dtype=torch.uint8)
    dx = torch.empty_like(x)
    n = x.numel()
    t0 = timeit(lambda: ops.bn_act_fwd(x, C, scale, shift, 1, None, out=out))
    t1 = timeit(lambda: ops.bn_act_fwd(x, C, scale, shift, 1, None, out=out, planes=pl, bn_bound=aux[0:1]))
    lib = ops._lib.load()
    def red_only():
        red.zero_()
        ops.check(lib.yolo_bn_act_bwd_reduce_bound(ops._p(x), ops._p(dout), P, C, ops._p(scale), ops._p(shift), ops._p(smean),
                                                   ops._p(sinv), 1, ops._p(red), ops._p(aux[1:69]), ops._stream()), "r")
    t2 = timeit(red_only)
    def app(dxo, plo):
        ops.check(lib.yolo_bn_act_bwd_apply_planes(ops._p(x), ops._p(dout), P, C, ops._p(g), ops._p(scale), ops._p(shift),
                                                   ops._p(smean), ops._p(sinv), 1, ops._p(red), None, None, ops._p(dxo),
                                                   ops._p(plo), ops._p(aux[1:69]), ops._stream()), "a")
    t3 = timeit(lambda: app(dx, None))
    t4 = timeit(lambda: app(None, pl))
    print(f"{h:4d} {C:4d} | {t0:6.3f} {n*8/t0/1e6:5.0f} | {t1:6.3f} {n*12/t1/1e6:5.0f} | {t2:6.3f} {n*8/t2/1e6:5.0f} | "
          f"{t3:6.3f} {n*12/t3/1e6:5.0f} | {t4:6.3f} {n*12/t4/1e6:5.0f}", flush=True)
