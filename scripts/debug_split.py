import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tf2_yolo_amd import ops
torch.manual_seed(0)
N, H, C, K = 1, 16, 32, 64
d = ops.conv_desc((N, H, H, C), K, 1, 1, 1, "same")
def run(x, w):
    y = ops.conv2d_fwd(d, x.cuda().float().contiguous(), w.cuda().float().contiguous())
    ref = torch.einsum("nhwc,kc->nhwk", x.double(), w.double())
    return ((y.cpu().double() - ref).abs().max() / ref.abs().max()).item()
x = torch.randn(N, H, H, C); w = torch.randn(K, 1, 1, C).reshape(K, C)
xq = x.bfloat16().float(); wq = w.bfloat16().float()
print("bf16-exact inputs (only h planes non-zero):", run(xq, wq))
print("x full, w bf16-exact:", run(x, wq))
print("x bf16-exact, w full:", run(xq, w))
print("both full:", run(x, w))
# which k positions are wrong? one-hot channel tests
for c in (0, 1, 3, 4, 7, 8, 15, 16, 31):
    xo = torch.zeros(N, H, H, C); xo[..., c] = x[..., c]
    print("only channel", c, run(xo, w))
