import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from tf2_yolo_amd import ops
torch.manual_seed(0)
N, H, C, K = 1, int(os.environ.get("DBG_H", "8")), int(os.environ.get("DBG_C", "32")), int(os.environ.get("DBG_K", "64"))
d = ops.conv_desc((N, H, H, C), K, 1, 1, 1, "same")
x = torch.randint(-3, 4, (N, H, H, C)).float(); w = torch.randint(-3, 4, (K, C)).float()
y = ops.conv2d_fwd(d, x.cuda().contiguous(), w.cuda().contiguous()).cpu()
ref = torch.einsum("nhwc,kc->nhwk", x, w)
diff = (y - ref).reshape(-1, K)
print("max diff", diff.abs().max().item())
bad = (diff.abs() > 1e-3)
print("bad rows:", bad.any(1).nonzero().flatten().tolist()[:64])
print("bad cols:", bad.any(0).nonzero().flatten().tolist()[:64])
print("y[0,:8]", y.reshape(-1, K)[0, :8].tolist()); print("ref[0,:8]", ref.reshape(-1, K)[0, :8].tolist())
# identity-like test: x one-hot pixel p channel c -> y[p,k] = w[k,c]
xo = torch.zeros(N, H, H, C); xo.reshape(-1, C)[5, 3] = 1.0
yo = ops.conv2d_fwd(d, xo.cuda().contiguous(), w.cuda().contiguous()).cpu().reshape(-1, K)
print("one-hot: nonzero rows", (yo.abs().sum(1) > 0).nonzero().flatten().tolist(), "row5 == w[:,3]?", torch.equal(yo[5], w[:, 3]))
print(yo[5, :16].tolist()); print(w[:16, 3].tolist())
