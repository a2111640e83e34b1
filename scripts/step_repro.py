"""Is the training step bit-reproducible? Two models of one configuration built from the same seed take the same K steps on
the same batch (one after the other, same process); after every step their parameters and gradients are compared bitwise and
the first tensors that differ are named. usage: step_repro.py c1|c2|c3|c4 [steps] [batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
import numpy as np, torch
from tf2_yolo_amd import graphs, labels, optimizers, ops
ops.create_side_streams()
cfg = sys.argv[1] if len(sys.argv) > 1 else "c2"
K = int(sys.argv[2]) if len(sys.argv) > 2 else 6


def make():
    if cfg == "c1":
        import yolov1_5
        y = yolov1_5.Yolo((224, 224, 3), ["raccoon"]); y.create_model(bbox_num=2)
        return y, y.loss(binary_weight=0.5), 4, 1, 56
    if cfg == "c2":
        import yolov2
        y = yolov2.Yolo((416, 416, 3), [f"c{i}" for i in range(20)]); y.create_model()
        return y, y.loss(), 16, 1, 32
    if cfg == "c4":
        import yolov4
        y = yolov4.Yolo((608, 608, 3), [f"c{i}" for i in range(80)])
        y.create_model(anchors=graphs.V4_DEFAULT_ANCHORS, pretrained_body=None)
        return y, y.loss(), 16, 3, 8
    import yolov3
    y = yolov3.Yolo((416, 416, 3), [f"c{i}" for i in range(80)]); y.create_model(pretrained_body=None, seed=1234)
    return y, y.loss(), 32, 3, 8


runs = []
for tag in ("A", "B"):
    if os.environ.get("REPRO_SEED", "1") != "0":
        torch.manual_seed(1234); np.random.seed(1234)
    y, loss, bs, levels, stride = make()
    if len(sys.argv) > 3:
        bs = int(sys.argv[3])
    m = y.model
    m.compile(optimizer=optimizers.Adam(learning_rate=1e-4), loss=loss)
    H = y.input_shape[0]
    x, ys = labels.synthetic_batch(np.random.default_rng(1234), bs, (H, H), y.class_num, levels=levels, finest_stride=stride)
    x = torch.from_numpy(x).cuda(); ys = [torch.from_numpy(t).cuda() for t in ys]
    snaps = [(m.net.params.data.clone(), None)]
    for i in range(K):
        bufs, _ = m.train_step_device(x, ys)
        if os.environ.get("REPRO_SYNC", "1") != "0":
            torch.cuda.synchronize()   # (REPRO_SYNC=0: the host runs ahead as in a benchmark loop; the clones below are stream-ordered)
        snaps.append((m.net.params.data.clone(), m.net.grads.clone(), [b[0].clone() for b in bufs]))
    runs.append((snaps, m))
net = runs[0][1].net
import hashlib
for tag, (snaps, m) in zip("AB", runs):
    print(cfg, tag, "sha1 of the parameters: initial", hashlib.sha1(snaps[0][0].cpu().numpy().tobytes()).hexdigest()[:12],
          "after", K, "steps", hashlib.sha1(snaps[K][0].cpu().numpy().tobytes()).hexdigest()[:12])
print(cfg, "initial parameters identical:", bool(torch.equal(runs[0][0][0][0], runs[1][0][0][0])))
for i in range(1, K + 1):
    pa, ga, la = runs[0][0][i]; pb, gb, lb = runs[1][0][i]
    la, lb = [float(t.item()) for t in la], [float(t.item()) for t in lb]
    dp, dg = (pa != pb), (ga != gb)
    print(f"step {i}: params differing {int(dp.sum())}, grads differing {int(dg.sum())}, losses {la} / {lb}")
    if int(dp.sum()):   # (the gradients are zeroed by the optimizer's launch: the parameters carry the difference)
        names = []
        for name in net.params.order:
            s = net.params.specs[name]
            k = int(dp[s.offset:s.offset + s.size].sum().item())
            if k:
                d = (pa[s.offset:s.offset + s.size].double() - pb[s.offset:s.offset + s.size].double()).abs().max().item()
                names.append((name, k, s.size, d))
        print("   parameter tensors that differ (name, elements, size, max |diff|):", names[:12], "... of", len(names))
        break
