"""Device time of the fused loss kernel on the benchmark's levels (C3: bs 32, C = 80, grids 13 / 26 / 52; C4: bs 16, 19 / 38 / 76)
and its compulsory bytes (y_true + y_pred read, gradient written). usage: python scripts/loss_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tf2_yolo_amd import graphs, labels, ops

for ver, N, hw, grids, anchors in ((3, 32, 416, (13, 26, 52), graphs.V3_DEFAULT_ANCHORS), (4, 16, 608, (19, 38, 76), graphs.V4_DEFAULT_ANCHORS)):
    rng = np.random.default_rng(1)
    _, ys = labels.synthetic_batch(rng, N, (hw, hw), 80)
    for i, g in enumerate(grids):
        cfg = ops.make_loss_cfg(ver, N, g, g, 3, 80, anchors=anchors[3 * i:3 * i + 3],
                                loss_weight=(1, 1, 5, 1) if ver == 3 else (1, 5, 1))
        yt = torch.from_numpy(ys[i]).cuda()
        yp = torch.rand(N, g, g, 255, device="cuda") * 0.98 + 0.01
        dp = torch.empty_like(yp)
        out = torch.zeros(8, device="cuda", dtype=torch.float64)
        res = {}
        for name, opt in (("chunk-ahead loader (round 4)", 8), ("cell-ahead loader (round 6)", 0)):
            ops.set_option(8, opt)    # OPT_EXP bit 8: the old loader
            for _ in range(3):
                ops.loss_fwd_bwd(cfg, yt, yp, loss_out=out, dpred=dp)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.loss_fwd_bwd(cfg, yt, yp, loss_out=out, dpred=dp)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 20 * 1e3
            mb = (yt.numel() + 2 * yp.numel()) * 4 / 1e6
            res[opt] = (dp.clone(), float(out[0]))
            print(f"v{ver} N={N} grid {g:3d} {name}: {us:7.1f} us per call (incl. the 64-byte memset), {mb:6.1f} MB compulsory, "
                  f"{mb / us:5.2f} TB/s, loss {float(out[0]):.6f}")
        ops.reset_options()
        print("    same gradient bit for bit:", bool(torch.equal(res[0][0], res[8][0])), " loss rel diff", abs(res[0][1] - res[8][1]) / abs(res[8][1]))
