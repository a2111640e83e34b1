"""Where does the device's loss gradient differ from the oracle's AT THE SAME predictions? (debug aid)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_model as T
version, hw, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
y, model, fwd, loss_o, loss_g, x, ys = T._setup(version, hw=hw, N=N)
net = model.net
outs = net.forward(torch.tensor(x).cuda(), training=True)
for lvl, (lf, lo, o, yt) in enumerate(zip(loss_g, loss_o, outs, ys)):
    dec = torch.zeros((o.shape[0] * o.shape[1] * o.shape[2], 2), dtype=torch.int32, device="cuda")
    lv, dp = lf.fwd_bwd(torch.tensor(yt).cuda(), o, decisions=dec)
    p64 = o.detach().double().cpu().requires_grad_(True)
    st = {}
    l = lo(torch.tensor(yt, dtype=torch.float64), p64, decide_with=dec.cpu(), stats=st)
    l.backward()
    g_dev, g_ref = dp.double().cpu(), p64.grad
    diff = (g_dev - g_ref).abs()
    print(f"level {lvl}: loss dev {lv[0].item():.9f} oracle {l.item():.9f}; grad max|diff| {diff.max().item():.3e} rel to max|g| {diff.max().item() / g_ref.abs().max().item():.3e}; decisions disagree {st}")
    flat = diff.reshape(-1, diff.shape[-1])
    cell = int(flat.max(dim=1).values.argmax())
    ch = int(flat[cell].argmax())
    A, D = 3, g_ref.shape[-1] // 3
    print("   worst cell", cell, "channel", ch, "(anchor", ch // D, "k", ch % D, ") dev", g_dev.reshape(-1, g_dev.shape[-1])[cell, ch].item(),
          "ref", g_ref.reshape(-1, g_ref.shape[-1])[cell, ch].item(), "decision", dec.cpu()[cell].tolist())
    tt = torch.tensor(yt).reshape(-1, yt.shape[-1])[cell]
    pp = o.detach().cpu().reshape(-1, o.shape[-1])[cell].reshape(A, D)
    print("   y_true[:5]", tt[:5].tolist())
    for a in range(A):
        print("   pred anchor", a, pp[a, :5].tolist())
