"""Per-unit activation error profile (GPU fp32 / CPU fp32 oracle vs fp64 oracle), training mode."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import test_gpu_model as T
version = int(sys.argv[1]) if len(sys.argv) > 1 else 4
y, model, fwd, loss_o, loss_g, x, ys = T._setup(version)
net = model.net
w = T._weights_dict(model)
outs = net.forward(torch.tensor(x).cuda(), training=True)
_, c64 = fwd({k: torch.tensor(v, dtype=torch.float64) for k, v in w.items()}, torch.tensor(x, dtype=torch.float64), True)
_, c32 = fwd({k: torch.tensor(v) for k, v in w.items()}, torch.tensor(x), True)
for u in net.units:
    if u.kind != "conv" or u.residual is not None:
        continue
    a = net.act[u.out.tid].cpu().double().numpy()
    r = c64.acts[u.name].numpy()
    r32 = c32.acts[u.name].double().numpy()
    print(f"{u.name:26s} {str(tuple(a.shape)):20s} gpu {T._rel(a, r):.2e} cpu32 {T._rel(r32, r):.2e}  |a|max {np.abs(r).max():.2e} std {r.std():.2e}")
