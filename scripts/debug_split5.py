import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tf2_yolo_amd import ops
torch.manual_seed(0)
N, H, C, K = 1, 8, 32, 64
d = ops.conv_desc((N, H, H, C), K, 1, 1, 1, "same")
for top in (3, 7, 15, 31, 63, 127, 128, 129, 255):
    g = lambda s: torch.randint(-top, top + 1, s).float()
    x = g((N, H, H, C)); w = g((K, C))
    y = ops.conv2d_fwd(d, x.cuda().contiguous(), w.cuda().contiguous()).cpu().double()
    ref = torch.einsum("nhwc,kc->nhwk", x.double(), w.double())
    err = (y - ref).abs()
    print(f"ints <= {top:4d}: max abs err {err.max().item():10.1f}  frac wrong {(err > 0).float().mean().item():.3f}")
# a single nonzero product: x[p,c]=a, w[k,c]=b
for a_, b_ in ((129.0, 1.0), (1.0, 129.0), (255.0, 255.0), (131.0, 3.0)):
    x = torch.zeros(N, H, H, C); w = torch.zeros(K, C)
    x.reshape(-1, C)[3, 20] = a_; w[7, 20] = b_
    y = ops.conv2d_fwd(d, x.cuda().contiguous(), w.cuda().contiguous()).cpu().reshape(-1, K)
    print(f"single product {a_} x {b_} (channel 20, 2nd stage): got {y[3,7].item()}")
    x = torch.zeros(N, H, H, C); w = torch.zeros(K, C)
    x.reshape(-1, C)[3, 4] = a_; w[7, 4] = b_
    y = ops.conv2d_fwd(d, x.cuda().contiguous(), w.cuda().contiguous()).cpu().reshape(-1, K)
    print(f"single product {a_} x {b_} (channel 4, 1st stage): got {y[3,7].item()}")
