"""Streaming rates of the BatchNorm kernels on YOLOv3-416 bs-32 layer shapes (torch events, 20 launches each).
usage: python scripts/bn_bench.py   (YOLO_BN_REDUCE_BIG=0 selects the 256-thread reduce everywhere)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf2_yolo_amd import ops, _lib
from tf2_yolo_amd.ops import _p, _stream

SHAPES = [(32 * 208 * 208, 64), (32 * 104 * 104, 128), (32 * 52 * 52, 256), (32 * 26 * 26, 512), (32 * 13 * 13, 1024),
          (32 * 52 * 52, 128), (32 * 416 * 416, 32)]
if os.environ.get("BN_BENCH_SHAPES") == "c4":   # YOLOv4-608 bs 16: CSP stages (Mish) and neck (LeakyReLU)
    SHAPES = [(16 * 304 * 304, 64), (16 * 152 * 152, 64), (16 * 152 * 152, 128), (16 * 76 * 76, 128), (16 * 76 * 76, 256),
              (16 * 38 * 38, 256), (16 * 38 * 38, 512), (16 * 19 * 19, 512), (16 * 19 * 19, 1024)]


def timeit(f, iters=20):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    lib = _lib.load()
    ACT = int(os.environ.get("BN_BENCH_ACT", "1"))    # 1 = LeakyReLU, 2 = Mish
    for P, C in SHAPES:
        x = torch.randn(P, C, device="cuda")
        dout = torch.randn(P, C, device="cuda")
        scale, shift = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda") * 0.1
        mean, inv = torch.randn(C, device="cuda") * 0.1, torch.rand(C, device="cuda") + 0.5
        red = torch.zeros((ops.BN_RED_SLOTS + 1) * 2 * C, device="cuda", dtype=torch.float64)
        aux = torch.zeros(72, device="cuda", dtype=torch.int32)
        planes = torch.zeros(ops.planes_bytes(P, C), device="cuda", dtype=torch.uint8)
        bound = torch.zeros(1, device="cuda", dtype=torch.int32)
        dg, db = torch.zeros(C, device="cuda"), torch.zeros(C, device="cuda")
        n = P * C

        def reduce():
            ops.check(lib.yolo_bn_act_bwd_reduce_bound(_p(x), _p(dout), P, C, _p(scale), _p(shift), _p(mean), _p(inv), ACT,
                                                       _p(red), _p(aux), _stream()), "reduce")

        def apply():
            ops.check(lib.yolo_bn_act_bwd_apply_planes(_p(x), _p(dout), P, C, _p(scale), _p(scale), _p(shift), _p(mean), _p(inv),
                                                       ACT, _p(red), _p(dg), _p(db), None, _p(planes), _p(aux), _stream()), "apply")

        def fwd():
            bound.fill_(0x40800000)
            ops.bn_act_fwd(x, C, scale, shift, ACT, planes=planes, want_out=False, bn_bound=bound)

        t_r, t_a, t_f = timeit(reduce), timeit(apply), timeit(fwd)
        print(f"P={P:8d} C={C:5d}  reduce+sum {t_r*1e6:7.1f} us {8*n/t_r/1e12:5.2f} TB/s | apply(planes) {t_a*1e6:7.1f} us "
              f"{12*n/t_a/1e12:5.2f} TB/s | fwd(planes) {t_f*1e6:7.1f} us {8*n/t_f/1e12:5.2f} TB/s", flush=True)


if __name__ == "__main__":
    main()
