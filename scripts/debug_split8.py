import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tf2_yolo_amd import ops
torch.manual_seed(0)
N, H, K, C = 1, 8, 64, 32
d = ops.conv_desc((N, H, H, C), K, 1, 1, 1, "same")
big = lambda s: torch.randint(-255, 256, s).float()
sm = lambda s: torch.randint(-3, 4, s).float()
for lo, hi in ((0, 16), (16, 32), (0, 32)):
    x = torch.zeros(N, H, H, C); x[..., lo:hi] = big((N, H, H, hi - lo)); w = torch.ones(K, C)
    y = ops.conv2d_fwd(d, x.cuda().contiguous(), w.cuda().contiguous()).cpu().double().reshape(-1, K)
    ref = torch.einsum("nhwc,kc->nhwk", x.double(), w.double()).reshape(-1, K)
    diff = y - ref
    print(f"x big only in channels {lo}:{hi}, w = 1: wrong {(diff != 0).float().mean().item():.3f} sample {diff[diff != 0][:6].tolist()}")
# one big element per row in channel c, rest zero, w = 1: recovers the element as seen by the kernel
for c in (0, 5, 15, 16, 21, 31):
    x = torch.zeros(N, H, H, C); vals = torch.arange(64).float() * 2 + 129; x.reshape(-1, C)[:, c] = vals
    y = ops.conv2d_fwd(d, x.cuda().contiguous(), torch.ones(K, C).cuda()).cpu().reshape(-1, K)
    print(f"channel {c}: kernel sees {y[:6, 0].tolist()} expected {vals[:6].tolist()}")
# dense in stage 1 but x = 129 everywhere
x = torch.zeros(N, H, H, C); x[..., 16:32] = 129.0
y = ops.conv2d_fwd(d, x.cuda().contiguous(), torch.ones(K, C).cuda()).cpu().reshape(-1, K)
print("all 129 in channels 16:32 ->", y[0, :4].tolist(), "expected", 129.0 * 16)
x = torch.zeros(N, H, H, C); x[..., 0:16] = 129.0
y = ops.conv2d_fwd(d, x.cuda().contiguous(), torch.ones(K, C).cuda()).cpu().reshape(-1, K)
print("all 129 in channels 0:16 ->", y[0, :4].tolist(), "expected", 129.0 * 16)
