"""One YOLOv3-416 training step (forward, loss, backward) at a given batch; prints a JSON line with the loss parts
and gradient statistics. Run it with YOLO_CONV_PLANES=1/0 (fp16 x 3 planes kernels vs exact bf16 x 6 kernels) or
YOLO_CONV_MODE=fp32 to compare the arithmetic paths at full layer sizes.
usage: python scripts/full_size_step.py [batch]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import yolov3
from tf2_yolo_amd import labels, optimizers

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
yolo = yolov3.Yolo((416, 416, 3), [f"c{i}" for i in range(80)])
yolo.create_model(pretrained_body=None, seed=1234)
m = yolo.model
m.compile(optimizer=optimizers.SGD(learning_rate=0.0), loss=yolo.loss())
rng = np.random.default_rng(99)
x_h, ys_h = labels.synthetic_batch(rng, N, (416, 416), 80)
x = torch.from_numpy(x_h).cuda()
ys = [torch.from_numpy(v).cuda() for v in ys_h]
net = m.net
outs = net.forward(x, training=True)
bufs = [torch.zeros(8, device="cuda", dtype=torch.float64) for _ in outs]
dp = [torch.empty_like(o) for o in outs]
for i, (o, yt) in enumerate(zip(outs, ys)):
    m.loss[i].fwd_bwd(yt, o, grad_scale=1.0, dpred=dp[i], loss_out=bufs[i])
net.backward(dp)
torch.cuda.synchronize()
g = net.grads.double()
print(json.dumps({"loss": [float(b[0].item()) for b in bufs],
                  "out_abs_sum": [float(o.double().abs().sum().item()) for o in outs],
                  "grad_l2": float(g.norm().item()), "grad_abs_sum": float(g.abs().sum().item()),
                  "grad_finite": bool(torch.isfinite(g).all().item())}))
