import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tf2_yolo_amd import ops
torch.manual_seed(0)
N, H, C, K = 1, int(os.environ.get("DBG_H", "8")), int(os.environ.get("DBG_C", "16")), 64
d = ops.conv_desc((N, H, H, C), K, 1, 1, 1, "same")
for name, gen in (("int", lambda s: torch.randint(-3, 4, s).float()),
                  ("half-int", lambda s: torch.randint(-7, 8, s).float() * 0.5),
                  ("bf16 rand", lambda s: torch.randn(s).bfloat16().float()),
                  ("pow2", lambda s: 2.0 ** torch.randint(-3, 4, s).float()),
                  ("1.5*pow2", lambda s: 1.5 * 2.0 ** torch.randint(-3, 4, s).float()),
                  ("neg pow2", lambda s: -(2.0 ** torch.randint(-3, 4, s).float())),
                  ("small ints/128", lambda s: torch.randint(-127, 128, s).float() / 128)):
    x = gen((N, H, H, C)); w = gen((K, C))
    y = ops.conv2d_fwd(d, x.cuda().contiguous(), w.cuda().contiguous()).cpu().double()
    ref = torch.einsum("nhwc,kc->nhwk", x.double(), w.double())
    print(f"{name:16s} rel err {((y-ref).abs().max()/ref.abs().max()).item():.3e}")
