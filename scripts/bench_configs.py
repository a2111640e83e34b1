"""Secondary configurations of BASELINE.json (configs[0], [1], [3], [4]); the headline config[2] is bench.py.
Prints one JSON line per measurement. usage: python scripts/bench_configs.py [c1] [c2] [c4] [c5]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from tf2_yolo_amd import graphs, labels, optimizers, tools

which = set(sys.argv[1:]) or {"c1", "c2", "c4", "c5"}


def train_bench(name, yolo, loss, batch, levels, finest_stride, steps=8, warmup=3):
    warmup = max(warmup, 3)   # the step is captured into hipGraphs during the third call (tf2_yolo_amd/capture.py)
    m = yolo.model
    m.compile(optimizer=optimizers.Adam(learning_rate=1e-4), loss=loss)
    rng = np.random.default_rng(1234)
    H = yolo.input_shape[0]
    x, ys = labels.synthetic_batch(rng, batch, (H, H), yolo.class_num, levels=levels, finest_stride=finest_stride)
    x = torch.from_numpy(x).cuda()
    ys = [torch.from_numpy(y).cuda() for y in ys]
    for _ in range(warmup):
        m.train_step_device(x, ys)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        bufs, _ = m.train_step_device(x, ys)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    loss_v = float(sum(b[0].item() for b in bufs))
    print(json.dumps({"config": name, "batch": batch, "ms_per_step": round(dt * 1e3, 3),
                      "launch_mode": type(m._step_graphs).__name__ if getattr(m, "_step_graphs", None) is not None else "eager",
                      "images_per_s": round(batch / dt, 2), "loss": round(loss_v, 4),
                      "finite": bool(np.isfinite(loss_v))}), flush=True)


if "c1" in which:
    import yolov1_5
    y = yolov1_5.Yolo((224, 224, 3), ["raccoon"])
    y.create_model(bbox_num=2)
    assert tuple(y.grid_shape) == (4, 4)
    # v1.5 labels: one level on the 4x4 grid (stride 56)
    train_bench("C1 YOLOv1.5 224x224 C=1 B=2 bs=4 (train step)", y, y.loss(binary_weight=0.5), 4, 1, 56)
    del y
if "c2" in which:
    import yolov2
    y = yolov2.Yolo((416, 416, 3), [f"c{i}" for i in range(20)])
    y.create_model()
    train_bench("C2 YOLOv2 Darknet-19 416x416 C=20 5 anchors bs=16 (train step)", y, y.loss(), 16, 1, 32)
    del y
if "c4" in which:
    import yolov4
    y = yolov4.Yolo((608, 608, 3), [f"c{i}" for i in range(80)])
    y.create_model(anchors=graphs.V4_DEFAULT_ANCHORS, pretrained_body=None)
    train_bench("C4 YOLOv4 CSPDarknet53+SPP+PAN 608x608 C=80 bs=16 (train step)", y, y.loss(), 16, 3, 8, steps=5, warmup=2)
    del y
if "c5" in which:
    import yolov3
    torch.cuda.empty_cache()
    y = yolov3.Yolo((416, 416, 3), [f"c{i}" for i in range(80)])
    y.create_model(pretrained_body=None)
    m = y.model
    rng = np.random.default_rng(1234)
    x = torch.from_numpy(rng.random((1, 416, 416, 3), dtype=np.float32)).cuda()

    def timed(fn, n=20):
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3, r

    t_fwd, outs = timed(lambda: m.net.forward(x, training=False))
    print(json.dumps({"config": "C5 YOLOv3 416 bs=1 inference forward (random weights), launches enqueued one by one",
                      "ms": round(t_fwd, 3)}), flush=True)
    t_g, outs_g = timed(lambda: m.net.infer(x))
    same = all(torch.equal(a, b) for a, b in zip(outs_g, m.net.forward(x, training=False)))
    print(json.dumps({"config": "C5 YOLOv3 416 bs=1 inference forward, hipGraph replay (Model.predict)", "ms": round(t_g, 3),
                      "same_as_eager": bool(same)}), flush=True)
    # decode + NMS on the model's own predictions (random weights) and on the BASELINE.md noise inputs
    lv = [outs[2][0], outs[1][0], outs[0][0]]
    t_dec, dec = timed(lambda: tools.decode_device(*lv, class_num=80, threshold=0.5, version=3))
    print(json.dumps({"config": "C5 decode on model output thr=.5", "rows": int(dec.shape[0]), "ms": round(t_dec, 3)}), flush=True)
    nrng = np.random.default_rng(1234)     # ONE generator, levels drawn 13 -> 26 -> 52: BASELINE.md's inputs (131 304 / 4 425 candidates)
    noise = [torch.from_numpy(nrng.random((g, g, 255), dtype=np.float32)).cuda() for g in (13, 26, 52)]
    for thr in (0.9, 0.5):
        t_dec, dec = timed(lambda: tools.decode_device(*noise, class_num=80, threshold=thr, version=3), n=10)
        res = {"config": f"C5 uniform-noise levels (13,26,52) thr={thr}", "rows": int(dec.shape[0]), "decode_ms": round(t_dec, 3)}
        for name, fn in (("nms", lambda: tools.nms(dec, class_num=80, nms_threshold=0.5)),
                         ("diou_nms", lambda: tools.nms(dec, class_num=80, nms_threshold=0.5, iou_mode=2)),
                         ("soft_nms", lambda: tools.soft_nms(dec, class_num=80, nms_threshold=0.5, conf_threshold=thr, sigma=0.5))):
            t, out = timed(fn, n=5)
            res[name + "_ms"] = round(t, 3)
            res[name + "_kept"] = int(out.shape[0])
        print(json.dumps(res), flush=True)
