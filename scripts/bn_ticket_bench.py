"""Times conv+stats (+ bn_finalize) against yolo_conv2d_fwd_planes_bn on a few layer shapes (torch events, launches
back to back on one stream). Usage: python scripts/bn_ticket_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tf2_yolo_amd import ops

SHAPES = [(32, 52, 128, 256, 3), (32, 26, 256, 512, 3), (32, 13, 512, 1024, 3), (32, 52, 256, 128, 1), (32, 13, 1024, 512, 1),
          (32, 104, 64, 128, 3)]


def main():
    ops.ensure_conv_workspace()
    for (n, h, cin, cout, k) in SHAPES:
        d = ops.conv_desc((n, h, h, cin), cout, k, k, 1, "same")
        x = torch.randn(n * h * h, cin, device="cuda")
        w = torch.randn(cout, k * k * cin, device="cuda") / (k * k * cin) ** 0.5
        xp, wp = ops.split_planes(x, n * h * h, cin), ops.split_planes(w, cout, k * k * cin)
        y = torch.empty(n, h, h, cout, device="cuda")
        st = torch.zeros(ops.BN_STAT_SLOTS * 2 * cout, device="cuda", dtype=torch.float64)
        amax = torch.zeros(cout, device="cuda", dtype=torch.int32)
        bound = torch.zeros(1, device="cuda", dtype=torch.int32)
        ticket = torch.zeros(1, device="cuda", dtype=torch.int32)
        gamma, beta = torch.ones(cout, device="cuda"), torch.zeros(cout, device="cuda")
        mm, mv = torch.zeros(cout, device="cuda"), torch.ones(cout, device="cuda")
        v = [torch.empty(cout, device="cuda") for _ in range(4)]
        P = n * h * h

        def conv_only():
            ops.conv2d_fwd_planes(d, xp, wp, None, out=y, stats=st, absmax=amax)

        def two():
            conv_only()
            ops.bn_finalize(st, P, cout, gamma, beta, mm, mv, *v, bound=bound, absmax=amax)

        def fused():
            ops.conv2d_fwd_planes_bn(d, xp, wp, None, y, st, amax, gamma, beta, mm, mv, *v, ticket, bound=bound)

        def timeit(f, iters=40):
            for _ in range(5):
                f()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                f()
            e1.record()
            torch.cuda.synchronize()
            return e0.elapsed_time(e1) / iters * 1e3

        res = {"conv": timeit(conv_only), "conv+finalize": timeit(two), "fused": timeit(fused)}
        print((n, h, cin, cout, k), " ".join(f"{k_}={v_:.1f}us" for k_, v_ in res.items()), flush=True)


if __name__ == "__main__":
    main()
