"""Where does the device's per-tensor gradient error stand against OTHER fp32 executions of the same problem?

VERDICT r02 weak #2: YOLOv4-608 at bs 1 failed `e < max(1e-3, 4 e32, 3 fwd_floor)` on one BN tensor per conv path
(3-5x the fp32-CPU error of that tensor). This script runs ONE problem (version, hw, N) and prints, per parameter
tensor, the error against the float64 oracle of
    gpu      the device (conv path = YOLO_CONV_MODE / YOLO_CONV_PLANES of this process)
    cpu32a   the float32 CPU execution of the oracle, all host threads
    cpu32b   the same float32 oracle on ONE thread (different blocking / summation order inside oneDNN)
and the distribution of the ratios gpu/cpu32a and cpu32b/cpu32a: if two fp32 CPU executions of one oracle scatter
by a factor of k per tensor, a per-tensor bound tighter than k says nothing about the device.
Usage: python scripts/grad_excess.py VERSION HW N [out.json]
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

import test_gpu_model as T

version, hw, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
out_path = sys.argv[4] if len(sys.argv) > 4 else None
y, model, fwd, loss_o, loss_g, x, ys = T._setup(version, hw=hw, N=N)
net = model.net
w = T._weights_dict(model)
from tf2_yolo_amd import optimizers

model.compile(optimizer=optimizers.Adam(1e-3), loss=loss_g)
outs = net.forward(torch.tensor(x).cuda(), training=True)
masks = T._gpu_leaky_masks(net)
decs = [torch.zeros((o.shape[0] * o.shape[1] * o.shape[2], 2), dtype=torch.int32, device="cuda") for o in outs]
dp = [lf.fwd_bwd(torch.tensor(yt).cuda(), o, decisions=dc)[1] for lf, o, yt, dc in zip(loss_g, outs, ys, decs)]
decs = [d.cpu() for d in decs]
net.backward(dp)
torch.cuda.synchronize()
g = net.grads.cpu().numpy()
dev_out = [o.cpu().numpy() for o in outs]


def oracle(dtype, threads=None):
    if threads:
        torch.set_num_threads(threads)
    wt = {k: torch.tensor(v, dtype=dtype, requires_grad=True) for k, v in w.items()}
    out, ctx = fwd(wt, torch.tensor(x, dtype=dtype), True, masks)
    # (loss gradient taken at the DEVICE's predictions, straight-through: see tests/test_gpu_model.py)
    tot = sum(lf(torch.tensor(yt, dtype=dtype), o + (torch.tensor(d, dtype=dtype) - o).detach(), decide_with=dc)
              for lf, yt, o, dc, d in zip(loss_o, ys, out, decs, dev_out))
    tot.backward()
    return wt, [o.detach().double().numpy() for o in out], ctx


nthreads = torch.get_num_threads()
w64, o64, c64 = oracle(torch.float64)
w32a, o32a, c32a = oracle(torch.float32)
w32b, o32b, c32b = oracle(torch.float32, threads=1)
torch.set_num_threads(nthreads)
print("forward outputs: gpu", [f"{T._rel(a, b):.2e}" for a, b in zip(dev_out, o64)],
      "cpu32a", [f"{T._rel(a, b):.2e}" for a, b in zip(o32a, o64)],
      "cpu32b", [f"{T._rel(a, b):.2e}" for a, b in zip(o32b, o64)])
rows = []
for n in model.layer_names():
    lw = model.get_layer(n).get_weights()
    if not lw or n.endswith("_anchor"):
        continue
    for i in range(len(lw)):
        r = w64[f"{n}/{i}"].grad
        if r is None:
            continue
        r = r.numpy()
        e_gpu = T._rel(T._grad_view(model, n, i, g), r)
        e_a = T._rel(w32a[f"{n}/{i}"].grad.numpy(), r)
        e_b = T._rel(w32b[f"{n}/{i}"].grad.numpy(), r)
        rows.append((n, i, e_gpu, e_a, e_b))
rows = [r for r in rows if r[3] > 0 and r[4] > 0]
rg = np.array([r[2] / r[3] for r in rows])
rb = np.array([r[4] / r[3] for r in rows])
q = [0.1, 0.5, 0.9, 0.99, 1.0]
print(f"{len(rows)} tensors; conv path: mode={os.environ.get('YOLO_CONV_MODE', 'split')} planes={os.environ.get('YOLO_CONV_PLANES', '1')}")
print("gpu/cpu32a   quantiles", dict(zip(q, np.round(np.quantile(rg, q), 2))), "count > 2:", int((rg > 2).sum()), "> 4:", int((rg > 4).sum()))
print("cpu32b/cpu32a quantiles", dict(zip(q, np.round(np.quantile(rb, q), 2))), "count > 2:", int((rb > 2).sum()), "> 4:", int((rb > 4).sum()))
print("max errors: gpu %.2e cpu32a %.2e cpu32b %.2e" % (max(r[2] for r in rows), max(r[3] for r in rows), max(r[4] for r in rows)))
print("worst ratio tensors (gpu/cpu32a):")
for k in np.argsort(-rg)[:12]:
    n, i, eg, ea, eb = rows[k]
    print(f"  {n:30s} {i} gpu {eg:.2e} cpu32a {ea:.2e} cpu32b {eb:.2e}")
if out_path:
    json.dump({"version": version, "hw": hw, "N": N, "conv_mode": os.environ.get("YOLO_CONV_MODE", "split"),
               "planes": os.environ.get("YOLO_CONV_PLANES", "1"),
               "rows": [{"layer": n, "i": i, "gpu": eg, "cpu32a": ea, "cpu32b": eb} for n, i, eg, ea, eb in rows]},
              open(out_path, "w"))
