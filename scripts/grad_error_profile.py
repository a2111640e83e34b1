"""Per-layer gradient error profile (GPU fp32 vs oracle fp64, and oracle fp32 vs oracle fp64)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import test_gpu_model as T

version = int(sys.argv[1]) if len(sys.argv) > 1 else 3
y, model, fwd, loss_o, loss_g, x, ys = T._setup(version)
net = model.net
w = T._weights_dict(model)

outs_by_dtype = {}
def oracle(dtype):
    wt = {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in w.items()}
    out, _ = fwd(wt, torch.tensor(x, dtype=dtype), True)
    outs_by_dtype[dtype] = [o.detach().double().numpy() for o in out]
    tot = sum(lf(torch.tensor(yt, dtype=dtype), o) for lf, yt, o in zip(loss_o, ys, out))
    tot.backward()
    return wt, tot.item()

w64, l64 = oracle(torch.float64)
w32, l32 = oracle(torch.float32)
from tf2_yolo_amd import optimizers
model.compile(optimizer=optimizers.Adam(1e-3), loss=loss_g)
outs = net.forward(torch.tensor(x).cuda(), training=True)
for a, b64, b32 in zip(outs, outs_by_dtype[torch.float64], outs_by_dtype[torch.float32]):
    print("fwd out: gpu", T._rel(a.cpu().numpy(), b64), "cpu32", T._rel(b32, b64))
dp = [lf.fwd_bwd(torch.tensor(yt).cuda(), o)[1] for lf, o, yt in zip(loss_g, outs, ys)]
net.backward(dp)
g = net.grads.cpu().numpy()
print("loss64", l64, "loss32", l32)
for n in model.layer_names():
    lw = model.get_layer(n).get_weights()
    if not lw or n.endswith("_anchor"): continue
    for i in range(len(lw)):
        r = w64[f"{n}/{i}"].grad
        if r is None: continue
        got = T._grad_view(model, n, i, g)
        e_gpu = T._rel(got, r.numpy())
        e_32 = T._rel(w32[f"{n}/{i}"].grad.numpy(), r.numpy())
        l_gpu = T._l2(got, r.numpy())
        l_32 = T._l2(w32[f"{n}/{i}"].grad.numpy(), r.numpy())
        print(f"{n:28s} {i} gpu {e_gpu:.2e}  cpu32 {e_32:.2e}  L2 gpu {l_gpu:.2e} cpu32 {l_32:.2e}")
