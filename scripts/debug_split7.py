import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tf2_yolo_amd import ops
torch.manual_seed(0)
N, H, K = 1, 8, 64
for C in (16, 32):
    d = ops.conv_desc((N, H, H, C), K, 1, 1, 1, "same")
    big = lambda s: torch.randint(-255, 256, s).float()
    sm = lambda s: torch.randint(-3, 4, s).float()
    for name, gx, gw in (("x big w small", big, sm), ("x small w big", sm, big), ("both big", big, big)):
        x = gx((N, H, H, C)); w = gw((K, C))
        y = ops.conv2d_fwd(d, x.cuda().contiguous(), w.cuda().contiguous()).cpu().double().reshape(-1, K)
        ref = torch.einsum("nhwc,kc->nhwk", x.double(), w.double()).reshape(-1, K)
        diff = y - ref
        bad = diff != 0
        print(f"C={C} {name}: wrong {bad.float().mean().item():.3f}; rows wrong {bad.any(1).sum().item()}/64 cols wrong {bad.any(0).sum().item()}/64; sample diffs {diff[bad][:8].tolist()}")
