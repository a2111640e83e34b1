"""eager vs eager vs captured training steps: where do they differ? (debug aid for tests/test_gpu_keras_shell.py)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import yolov3
from tf2_yolo_amd import labels
from tf2_yolo_amd.optimizers import Adam
A9 = [[0.89663461, 0.78365384], [0.375, 0.47596153], [0.27884615, 0.21634615], [0.14182692, 0.28605769],
      [0.14903846, 0.10817307], [0.07211538, 0.14663461], [0.07932692, 0.05528846], [0.03846153, 0.07211538],
      [0.02403846, 0.03125]]
HW = int(sys.argv[1]) if len(sys.argv) > 1 else 96
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4

def make(graphs):
    y = yolov3.Yolo((HW, HW, 3), list("abcdefgh"))
    y.create_model(anchors=A9, pretrained_body=None, seed=11)
    y.model.compile(optimizer=Adam(learning_rate=1e-3), loss=y.loss())
    y.model._graphs_failed = not graphs
    return y.model
rng = np.random.default_rng(3)
x1, ys1 = labels.synthetic_batch(rng, N, (HW, HW), 8)
b1 = (torch.from_numpy(x1).cuda(), [torch.from_numpy(a).cuda() for a in ys1])
runs = {}
for tag, graphs in (("eagerA", False), ("eagerB", False), ("graph", True)):
    m = make(graphs)
    losses, snaps = [], []
    for i in range(5):
        bufs, _ = m.train_step_device(*b1)
        losses.append([float(b[0].item()) for b in bufs])
        snaps.append((m.net.params.data.clone(), m.net.grads.clone()))
    runs[tag] = (losses, snaps, m)
for other in ("eagerB", "graph"):
    print("==", other, "vs eagerA")
    for i in range(5):
        la, lb = runs["eagerA"][0][i], runs[other][0][i]
        pa, pb = runs["eagerA"][1][i][0], runs[other][1][i][0]
        nd = int((pa != pb).sum().item())
        print(f" step {i}: loss diff {[abs(a - b) for a, b in zip(la, lb)]} params differing {nd}")
        if nd and i < 5:
            net = runs["eagerA"][2].net
            d = (pa != pb)
            names = []
            for name in net.params.order:
                s = net.params.specs[name]
                k = int(d[s.offset:s.offset + s.size].sum().item())
                if k:
                    names.append((name, k, s.size))
            print("   first differing tensors:", names[:6], "... last:", names[-3:])
            break
