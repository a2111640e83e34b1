import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tf2_yolo_amd import ops
torch.manual_seed(0)
N, H, K, C = 1, 8, 64, 32
d = ops.conv_desc((N, H, H, C), K, 1, 1, 1, "same")
x = torch.randint(-255, 256, (N, H, H, C)).float()
w = torch.zeros(K, C)
for k in range(K):
    w[k, k % 32] = 1.0
y = ops.conv2d_fwd(d, x.cuda().contiguous(), w.cuda().contiguous()).cpu().reshape(-1, K)
seen = y[:, :32]
xr = x.reshape(-1, C)
bad = (seen != xr)
print("elements misread:", bad.sum().item(), "of", bad.numel())
idx = bad.nonzero()[:12]
for p, c in idx.tolist():
    print(f"pixel {p} ch {c}: x = {xr[p, c].item()} seen {seen[p, c].item()}")
# now w = identity-like but with small random multipliers
w2 = torch.zeros(K, C)
mult = torch.randint(1, 4, (K,)).float()
for k in range(K):
    w2[k, k % 32] = mult[k]
y2 = ops.conv2d_fwd(d, x.cuda().contiguous(), w2.cuda().contiguous()).cpu().reshape(-1, K)
print("scaled one-hot wrong:", (y2[:, :32] != xr * mult[:32]).sum().item())
# dense w small, but x only one nonzero per row
w3 = torch.randint(-3, 4, (K, C)).float()
x3 = torch.zeros(64, C); x3[torch.arange(64), torch.arange(64) % 32] = xr[torch.arange(64), torch.arange(64) % 32]
y3 = ops.conv2d_fwd(d, x3.reshape(N, H, H, C).cuda().contiguous(), w3.cuda().contiguous()).cpu().reshape(-1, K)
print("one x per row, dense w wrong:", (y3 != x3 @ w3.T).sum().item())
y4 = ops.conv2d_fwd(d, x.cuda().contiguous(), w3.cuda().contiguous()).cpu().reshape(-1, K)
ref4 = xr @ w3.T
print("dense x dense w wrong:", (y4 != ref4).sum().item(), "of", ref4.numel(), " fp32 torch matmul exact?", torch.equal(ref4.double(), xr.double() @ w3.double().T))
