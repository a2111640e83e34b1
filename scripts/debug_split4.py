import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tf2_yolo_amd import ops
torch.manual_seed(0)
N, H, C, K = 1, 8, 32, 64
d = ops.conv_desc((N, H, H, C), K, 1, 1, 1, "same")
ints = lambda s: torch.randint(-3, 4, s).float()
big = lambda s: torch.randint(-255, 256, s).float()
bfr = lambda s: torch.randn(s).bfloat16().float()
bfu = lambda s: (torch.rand(s) + 1.0).bfloat16().float()          # [1,2): single binade
bfs = lambda s: (torch.randn(s) * 100).bfloat16().float()
for name, gx, gw in (("x bf16 rand, w int", bfr, ints), ("x int, w bf16 rand", ints, bfr), ("ints<=255 both", big, big),
                     ("bf16 in [1,2) both", bfu, bfu), ("bf16 rand*100", bfs, bfs)):
    x = gx((N, H, H, C)); w = gw((K, C))
    y = ops.conv2d_fwd(d, x.cuda().contiguous(), w.cuda().contiguous()).cpu().double()
    ref = torch.einsum("nhwc,kc->nhwk", x.double(), w.double())
    err = (y - ref).abs()
    i = err.argmax().item()
    print(f"{name:22s} rel err {(err.max()/ref.abs().max()).item():.3e}  worst idx {i} y {y.flatten()[i].item():.6f} ref {ref.flatten()[i].item():.6f}")
# stage isolation: zero out channels 0..15 or 16..31
for lo, hi in ((0, 16), (16, 32)):
    x = bfr((N, H, H, C)); w = bfr((K, C)); x[..., lo:hi] = 0
    y = ops.conv2d_fwd(d, x.cuda().contiguous(), w.cuda().contiguous()).cpu().double()
    ref = torch.einsum("nhwc,kc->nhwk", x.double(), w.double())
    print(f"channels {lo}-{hi} zero: rel err {((y-ref).abs().max()/ref.abs().max()).item():.3e}")
