"""The README's 6-step flow on the Keras-like shell (README.md:207-336 of the reference): compile with
loss + metrics, fit on arrays and on a Sequence-like, evaluate, predict, decode + nms."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

A9 = [[0.89663461, 0.78365384], [0.375, 0.47596153], [0.27884615, 0.21634615], [0.14182692, 0.28605769],
      [0.14903846, 0.10817307], [0.07211538, 0.14663461], [0.07932692, 0.05528846], [0.03846153, 0.07211538],
      [0.02403846, 0.03125]]


class _Seq:
    def __init__(self, x, ys, bs):
        self.x, self.ys, self.bs = x, ys, bs

    def __len__(self):
        return len(self.x) // self.bs

    def __getitem__(self, i):
        sl = slice(i * self.bs, (i + 1) * self.bs)
        return self.x[sl], [y[sl] for y in self.ys]


def test_readme_flow_v3():
    # the imports of /root/reference/README.md:207-336, verbatim
    from utils import tools
    from utils.tools import get_class_weight
    from utils.measurement import PR_func, create_score_mat  # noqa: F401
    from yolov3 import Yolo
    from tf2_yolo_amd import labels
    from tf2_yolo_amd.optimizers import Adam
    yolo = Yolo((64, 64, 3), ["a", "b"])
    yolo.create_model(anchors=A9, pretrained_body=None)
    rng = np.random.default_rng(0)
    x, ys = labels.synthetic_batch(rng, 8, (64, 64), 2)
    bw = [get_class_weight(y[..., 4:5], "binary")[0] for y in ys]     # README.md:229-233, 248-252
    yolo.model.compile(optimizer=Adam(lr=1e-4), loss=yolo.loss(bw), metrics=yolo.metrics("obj+iou+recall0.5"))
    h = yolo.model.fit(x, ys, batch_size=4, epochs=8, verbose=0)
    losses = h.history["loss"]
    # (Adam's first normalised steps overshoot on 4-image batches: compare the tail, not step 2, with the start)
    assert len(losses) == 8 and np.isfinite(losses).all() and min(losses[-2:]) < losses[0]
    yolo.model.fit(_Seq(x, ys, 4), epochs=1, verbose=0)
    ev = yolo.model.evaluate(x, ys, batch_size=4, verbose=0)
    # total, 3 level losses, 3x3 metrics. (After a dozen steps the moving statistics are still ~90 % their
    # initial 0/1 (momentum 0.99), so inference-mode losses of this random net may overflow, in Keras too;
    # the metrics are ratios of counts and must be proper fractions.)
    assert len(ev) == 1 + 3 + 9
    pred = yolo.model.predict(x, batch_size=4)
    assert [p.shape for p in pred] == [(8, 2, 2, 21), (8, 4, 4, 21), (8, 8, 8, 21)]
    boxes = tools.decode(pred[2][0], pred[1][0], pred[0][0], class_num=2, threshold=0.3, version=3)
    if boxes.size:
        kept = tools.nms(boxes, class_num=2, nms_threshold=0.5)
        assert kept.shape[1] == 7 and len(kept) <= len(boxes)
    # metric closures are callable like tf.keras metrics and agree with the evaluate() aggregation order
    m = yolo.metrics("obj+iou+class+recall0.6")
    assert [f.kind for f in m[0]] == ["obj_acc", "mean_iou", "class_acc", "recall"] and m[0][3].iou_threshold == 0.6
    out = yolo.model(x[:4], training=True)           # batch-statistics forward: well-scaled predictions
    v = float(m[2][0](ys[2][:4], out[2]))
    assert 0.0 <= v <= 1.0
    # loss closure call == fused kernel loss
    lf = yolo.loss()[0]
    a = float(lf(ys[0][:4], out[0]))
    b = float(lf.fwd_bwd(ys[0][:4], out[0])[0][0])
    assert abs(a - b) <= 1e-5 * max(1.0, abs(b))


@pytest.mark.parametrize("ver", [2, 1, 4])
def test_train_steps_reduce_loss(ver):
    from tf2_yolo_amd import labels
    from tf2_yolo_amd.optimizers import Adam
    rng = np.random.default_rng(ver)
    if ver == 2:
        import yolov2
        y = yolov2.Yolo((64, 64, 3), ["a", "b", "c"])
        y.create_model()
        x, ys = labels.synthetic_batch(rng, 4, (64, 64), 3, levels=1, finest_stride=32)
        loss, lab = y.loss(), ys[0]
    elif ver == 1:
        import yolov1_5
        y = yolov1_5.Yolo((128, 128, 3), ["a"])
        y.create_model()
        x, ys = labels.synthetic_batch(rng, 4, (128, 128), 1, levels=1, finest_stride=64)
        loss, lab = y.loss(binary_weight=0.5), ys[0]
    else:
        import yolov4
        y = yolov4.Yolo((64, 64, 3), ["a", "b"])
        y.create_model(anchors=A9, pretrained_body=None)
        x, ys = labels.synthetic_batch(rng, 4, (64, 64), 2)
        loss, lab = y.loss(), ys
        assert np.allclose(np.array(y.anchors), np.array(A9), atol=1e-7)
        y.reshape_anchors((128, 128))
        assert np.allclose(np.array(y.anchors), 2 * np.array(A9), atol=1e-6)
        y.anchors = A9
    # Adam's first normalised steps overshoot on these 4-image batches (the loss rises for 3-4 steps
    # whatever the conv arithmetic), so train past that and compare the tail with the start
    y.model.compile(optimizer=Adam(learning_rate=1e-4), loss=loss)
    hist = []
    for _ in range(14):
        l = y.model.train_on_batch(x, lab)
        hist.append(float(l[0] if isinstance(l, list) else l))
    first, last = hist[0], min(hist[-3:])
    assert np.isfinite(last) and last < first


def test_v4_trainable_anchors():
    """yolov4 `anchors_trainable` (yolov4/__init__.py:147-159): d(loss)/d(anchor) from the head backward kernel
    agrees with central differences of the loss, and an optimizer step moves the anchors (and only when asked)"""
    import torch
    import yolov4
    from tf2_yolo_amd import labels
    from tf2_yolo_amd.optimizers import SGD
    y = yolov4.Yolo((64, 64, 3), ["a", "b"])
    y.create_model(anchors=A9, pretrained_body=None)
    rng = np.random.default_rng(5)
    x_h, ys_h = labels.synthetic_batch(rng, 4, (64, 64), 2)
    x = torch.from_numpy(x_h).cuda()
    ys = [torch.from_numpy(v).cuda() for v in ys_h]
    m = y.model
    m.compile(optimizer=SGD(learning_rate=1e-3), loss=y.loss())
    net = m.net
    a0 = np.array(y.anchors)
    m.train_step_device(x, ys)
    assert np.allclose(np.array(y.anchors), a0)           # not trainable: untouched
    assert y.anchors_trainable is False
    y.anchors_trainable = True

    def total_loss():
        outs = net.forward(x, training=True)
        bufs = [torch.zeros(8, device="cuda", dtype=torch.float64) for _ in outs]
        dp = [torch.empty_like(o) for o in outs]
        for i, (o, yt) in enumerate(zip(outs, ys)):
            m.loss[i].fwd_bwd(yt, o, grad_scale=1.0, dpred=dp[i], loss_out=bufs[i])
        return sum(float(b[0].item()) for b in bufs), dp

    net.anchor_grads.zero_()
    _, dp = total_loss()
    net.backward(dp)
    g = net.anchor_grads.double().cpu().numpy().copy()
    net.grads.zero_(); net.anchor_grads.zero_()
    assert np.isfinite(g).all() and np.abs(g).max() > 0
    base = net.anchors_flat.clone()
    k = int(np.argmax(np.abs(g)))
    eps = 1e-3 * float(base[k])
    net.anchors_flat[k] = base[k] + eps
    lp, _ = total_loss()
    net.anchors_flat[k] = base[k] - eps
    lm, _ = total_loss()
    net.anchors_flat.copy_(base)
    fd = (lp - lm) / (2 * eps)
    assert abs(fd - g[k]) <= 2e-2 * max(abs(g[k]), abs(fd)), (fd, g[k])
    # one SGD step moves every anchor against its gradient
    m.train_step_device(x, ys)
    moved = net.anchors_flat.double().cpu().numpy() - base.double().cpu().numpy()
    assert np.abs(moved).max() > 0 and (np.sign(moved[np.abs(g) > 1e-6]) == -np.sign(g[np.abs(g) > 1e-6])).all()
    assert not np.allclose(np.array(y.anchors), a0)


def test_graph_inference_matches_eager_and_tracks_weight_updates():
    """Network.infer (hipGraph replay, used by Model.predict) == forward(training=False); a weight change drops the graph"""
    import torch
    import yolov3
    y = yolov3.Yolo((64, 64, 3), ["a", "b"])
    y.create_model(anchors=A9, pretrained_body=None)
    net = y.model.net
    rng = np.random.default_rng(0)
    x1 = torch.from_numpy(rng.random((2, 64, 64, 3), dtype=np.float32)).cuda()
    x2 = torch.from_numpy(rng.random((2, 64, 64, 3), dtype=np.float32)).cuda()
    for x in (x1, x2, x1):
        g = [o.clone() for o in net.infer(x)]
        e = [o.clone() for o in net.forward(x, training=False)]
        assert all(torch.equal(a, b) for a, b in zip(g, e))
    assert 2 in net._infer_graphs
    k = net.params.order[0]
    before = [o.clone() for o in net.infer(x1)]
    net.params.view(k).mul_(0.5)
    net.mark_params_changed()
    assert not net._infer_graphs
    g = [o.clone() for o in net.infer(x1)]
    e = [o.clone() for o in net.forward(x1, training=False)]
    # (an untrained net in inference mode may overflow to inf/nan, as in Keras: compare nan-aware)
    same = lambda a, b: bool(torch.isclose(a, b, rtol=0, atol=0, equal_nan=True).all())
    assert all(same(a, b) for a, b in zip(g, e))
    assert not all(same(a, b) for a, b in zip(g, before))          # the new weights are in the new graph
    p = y.model.predict(x1.cpu().numpy())
    assert all(np.array_equal(a, b.cpu().numpy(), equal_nan=True) for a, b in zip(p, e))


def test_fit_streams_host_batches_like_the_plain_loop(monkeypatch):
    """fit() on host arrays goes through pinned staging + a copy stream (feeder.HostFeeder) without per-batch
    synchronisation; the plain train_on_batch loop (YOLO_FIT_PIPELINE=0) is the reference behaviour. With a
    learning rate of 0 the weights stay put, so every epoch's mean loss is a pure function of WHICH rows formed
    WHICH batch (10 images in shuffled batches of 4, 4, 2: batch-statistics BN makes the grouping matter) and must
    agree between the two loops to rounding. (With lr > 0 this tiny net - BatchNorm over 8-16 samples - amplifies
    the rounding noise of the filter-gradient atomics to 1e-3 within three steps, in either loop alike.)"""
    import yolov3
    from tf2_yolo_amd import labels
    from tf2_yolo_amd.optimizers import SGD
    rng = np.random.default_rng(3)
    x, ys = labels.synthetic_batch(rng, 10, (64, 64), 2)
    hist = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("YOLO_FIT_PIPELINE", mode)
        yolo = yolov3.Yolo((64, 64, 3), ["a", "b"])
        yolo.create_model(anchors=A9, pretrained_body=None, seed=7)
        yolo.model.compile(optimizer=SGD(learning_rate=0.0), loss=yolo.loss())
        h = yolo.model.fit(x, ys, batch_size=4, epochs=3, verbose=0)
        h2 = yolo.model.fit(_Seq(x, ys, 5), epochs=1, verbose=0)
        hist[mode] = np.array(h.history["loss"] + h2.history["loss"])
    assert np.isfinite(hist["1"]).all()
    assert len(set(np.round(hist["1"][:3], 3))) == 3      # the three shuffles group the rows differently
    np.testing.assert_allclose(hist["1"], hist["0"], rtol=1e-6)


def test_feeder_delivers_every_row_once_in_order():
    from tf2_yolo_amd.feeder import HostFeeder
    n = 23
    x = np.arange(n * 6, dtype=np.float32).reshape(n, 2, 3, 1)
    y = [np.arange(n * 2, dtype=np.float64).reshape(n, 2), -np.arange(n, dtype=np.float32).reshape(n, 1)]
    order = np.random.default_rng(0).permutation(n)
    from tf2_yolo_amd.feeder import FeederBuffers
    shared = FeederBuffers()
    for epoch in range(3):   # the Model keeps one FeederBuffers: slots, device sets and events are reused
        order = np.random.default_rng(epoch).permutation(n)
        f = HostFeeder(("arrays", [x] + y, order, 5), shared)
        got_x, got_y0, got_y1 = [], [], []
        busy = torch.zeros(1 << 22, device="cuda")
        try:
            for xb, yb in f:
                assert xb.is_cuda and xb.dtype == torch.float32 and len(yb) == 2
                for _ in range(20):   # keep the compute stream behind the host, as a training step does
                    busy.add_(1.0)
                got_x.append(xb.clone()); got_y0.append(yb[0].clone()); got_y1.append(yb[1].clone())
        finally:
            f.close()
        assert [len(g) for g in got_x] == [5, 5, 5, 5, 3]
        np.testing.assert_array_equal(torch.cat(got_x).cpu().numpy(), x[order])
        np.testing.assert_array_equal(torch.cat(got_y0).cpu().numpy(), y[0][order].astype(np.float32))
        np.testing.assert_array_equal(torch.cat(got_y1).cpu().numpy(), y[1][order])
    # an exception inside the producer surfaces on the consumer's thread
    def bad():
        yield x[:2], [a[:2] for a in y]
        raise RuntimeError("boom")
    f = HostFeeder(("batches", bad()))
    with pytest.raises(RuntimeError, match="boom"):
        try:
            for _ in f:
                pass
        finally:
            f.close()


def test_utils_tools_cal_iou_bit_exact_vs_reference_goldens():
    """utils.tools.cal_iou (utils/tools.py:630-684) on the device against the reference's own outputs: the pairwise
    IoU / DIoU matrices of tests/golden/tools_golden.npz, bit for bit; float32 operands stay float32 like NumPy."""
    import os
    import sys
    from utils.tools import cal_iou
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    sys.path.insert(0, here)
    try:
        import gen_inputs
    finally:
        sys.path.remove(here)
    g = np.load(os.path.join(here, "tools_golden.npz"))
    boxes = gen_inputs.misc_inputs()["iou_boxes"]
    a, b = boxes.reshape(-1, 1, 5), boxes.reshape(1, -1, 5)
    iou = cal_iou(a, b, mode=1)
    assert iou.dtype == np.float64 and iou.shape == g["iou_mat"].shape
    assert np.array_equal(iou, g["iou_mat"])
    diou = cal_iou(a, b, mode=2)
    assert np.array_equal(diou, g["diou_mat"], equal_nan=True)
    # CUDA tensors in -> CUDA tensor out, same bits; the NMS call shape (expand_dims 0 / 1, utils/tools.py:712-715)
    t = torch.from_numpy(boxes).cuda()
    d = cal_iou(t[None, :, :4], t[:, None, :4], mode=1)
    assert d.is_cuda and np.array_equal(d.cpu().numpy(), g["iou_mat"].T)
    # float32 operands: NumPy computes in float32
    from oracle import tools as OT
    a32, b32 = a.astype(np.float32), b.astype(np.float32)
    r32 = cal_iou(a32, b32, mode=2)
    assert r32.dtype == np.float32 and np.array_equal(r32, OT.cal_iou(a32, b32, 2), equal_nan=True)
    assert cal_iou(a, b, mode=7) is None            # the reference falls through and returns None
    assert cal_iou(np.zeros((0, 4)), np.zeros((0, 4))).shape == (0,)


@pytest.mark.parametrize("ver", [3, 4])
def test_loss_side_cal_iou_vs_oracle(ver):
    """yolovN.losses.cal_iou (yolov3/losses/loss.py:9-37, yolov4/losses/loss.py:10-61) against the restatement the
    loss oracle uses, on the losses' own call shape (N,S,S,1,4) x (N,S,S,B,4). float32 like TF: 1e-6."""
    import importlib
    from oracle import losses as OL
    cal_iou = importlib.import_module(f"yolov{ver}.losses").cal_iou
    rng = np.random.default_rng(7)
    N, S, B = 2, 13, 3
    t = rng.random((N, S, S, 1, 4)).astype(np.float32)
    p = rng.random((N, S, S, B, 4)).astype(np.float32)
    t[..., 2:] = t[..., 2:] * 0.5 + 0.05
    p[..., 2:] = p[..., 2:] * 0.5 + 0.05
    t64, p64 = torch.tensor(t, dtype=torch.float64), torch.tensor(p, dtype=torch.float64)
    if ver == 4:
        iou, ciou = cal_iou(t, p, (S, S), return_ciou=True)
        ri, rc = OL.cal_iou(t64, p64, (S, S), return_ciou=True)
        assert np.abs(ciou.cpu().numpy() - rc.numpy()).max() <= 1e-6
    else:
        iou, ri = cal_iou(t, p, (S, S)), OL.cal_iou(t64, p64, (S, S))
    assert iou.shape == (N, S, S, B) and iou.dtype == torch.float32
    assert np.abs(iou.cpu().numpy() - ri.numpy()).max() <= 1e-6
    # a sliced view of a prediction tensor (the loss passes y_pred[..., :4]) needs no copy
    yp = torch.tensor(rng.random((N, S, S, B, 9)).astype(np.float32)).cuda()
    v = cal_iou(t, yp[..., :4], (S, S))
    v = v[0] if isinstance(v, tuple) else v
    ref = OL.cal_iou(t64, yp[..., :4].double().cpu(), (S, S))
    assert np.abs(v.cpu().numpy() - ref.numpy()).max() <= 1e-6


def test_yolo_body_plus_yolo_head_is_create_model():
    """the reference's two-call construction (yolov3/__init__.py:122-175): yolo_body(...) then yolo_head(...) gives the
    model create_model gives; body weights set on the symbolic body arrive in the model"""
    import yolov3
    from yolov3.models import yolo_body, yolo_head
    ref = yolov3.Yolo((64, 64, 3), ["a", "b"])
    ref.create_model(anchors=A9, pretrained_body=None, seed=5)
    body = yolo_body((64, 64, 3), pretrained_darknet=ref.model)
    assert body.output_shape == [(None, 2, 2, 1024), (None, 4, 4, 512), (None, 8, 8, 256)]
    m = yolo_head(body, class_num=2, anchors=A9, seed=99)
    assert [tuple(o.shape) for o in m.output] == [(None, 2, 2, 21), (None, 4, 4, 21), (None, 8, 8, 21)]
    for n in ("conv1_conv", "block3_4_3x3_bn", "last2_3_3x3_conv"):
        for a, b in zip(m.get_layer(n).get_weights(), ref.model.get_layer(n).get_weights()):
            assert np.array_equal(a, b)
    # head weights are the new model's own draw (seed 99), not the donor's
    assert not np.array_equal(m.get_layer("out1_box1_xy_conv").get_weights()[0],
                              ref.model.get_layer("out1_box1_xy_conv").get_weights()[0])
    # Keras-ordered weight list through BodyModel.set_weights
    body2 = yolo_body((64, 64, 3))
    names = body2.layer_names()
    body2.set_weights([w for n in names for w in ref.model.get_layer(n).get_weights()])
    m2 = yolo_head(body2, class_num=2, anchors=A9)
    assert np.array_equal(m2.get_layer("block5_4_3x3_conv").get_weights()[0],
                          ref.model.get_layer("block5_4_3x3_conv").get_weights()[0])
    x = np.random.default_rng(0).random((2, 64, 64, 3), dtype=np.float32)
    assert [p.shape for p in m2.predict(x)] == [(2, 2, 2, 21), (2, 4, 4, 21), (2, 8, 8, 21)]
    with pytest.raises(ValueError, match="multiple"):
        yolo_head(yolo_body((64, 64, 3)), class_num=2, anchors=A9[:8])


@pytest.mark.parametrize("mode", ["tape", "graph"])
def test_captured_step_is_bit_identical_to_eager_steps(mode, monkeypatch):
    """From the third step on train_step_device replays the recorded step: the launch tape (tape.py, the default) or
    captured hipGraphs (capture.py, YOLO_STEP_MODE=graph). Same seed, same data: six steps with the recording (two eager,
    the recording / capture, three replays) against six eager steps -- every loss equal (to the fp64 atomics' order,
    1e-12) and the weights, BatchNorm moving statistics and Adam moments BIT-identical (the filter-gradient reductions
    are atomics-free). A different batch goes through the same recording, and the learning rate changes on the way."""
    monkeypatch.setenv("YOLO_STEP_MODE", mode)
    import yolov3
    from tf2_yolo_amd import labels
    from tf2_yolo_amd.optimizers import Adam

    def make(graphs):
        # (8 classes: 39 head channels, so the head filter gradients run on the planes kernels like the 255-channel
        # heads of the benchmark; heads narrower than 32 channels fall to the fp32-input kernel, whose split-K still ends
        # in fp32 atomics)
        y = yolov3.Yolo((96, 96, 3), list("abcdefgh"))
        y.create_model(anchors=A9, pretrained_body=None, seed=11)
        y.model.compile(optimizer=Adam(learning_rate=1e-3), loss=y.loss())
        y.model._graphs_failed = not graphs        # eager reference: the switch a failed capture would flip
        return y.model
    rng = np.random.default_rng(3)
    x1, ys1 = labels.synthetic_batch(rng, 4, (96, 96), 8)
    x2, ys2 = labels.synthetic_batch(rng, 4, (96, 96), 8)
    dev = lambda x, ys: (torch.from_numpy(x).cuda(), [torch.from_numpy(a).cuda() for a in ys])
    b1, b2 = dev(x1, ys1), dev(x2, ys2)
    seq = [b1, b1, b1, b2, b1, b2]
    res = {}
    for graphs in (False, True):
        m = make(graphs)
        losses = []
        for i, (x, ys) in enumerate(seq):
            if i == 4:
                m.optimizer.learning_rate = 3e-4          # read by refresh_hyper / the eager step alike
            bufs, _ = m.train_step_device(x, ys)
            losses.append([float(b[0].item()) for b in bufs])
        if graphs:
            assert m._step_graphs is not None
            if mode == "graph":
                assert len(m._step_graphs.segments) == 1      # no data parallelism: one graph
            else:
                assert len(m._step_graphs.tape.entries) > 500
        res[graphs] = (losses, m.net.params.data.clone(), m.net.state.data.clone(), m.optimizer.m.clone(), m.optimizer.v.clone())
    for la, lb in zip(res[False][0], res[True][0]):
        for a, b in zip(la, lb):
            assert abs(a - b) <= 1e-12 * max(abs(a), 1.0), (res[False][0], res[True][0])
    for a, b in zip(res[False][1:], res[True][1:]):
        assert torch.equal(a, b)
    assert res[True][0][0] != res[True][0][3]       # (the second batch really went through the graphs)


def test_replayed_steps_do_not_depend_on_how_far_the_host_runs_ahead():
    """A loop that does not synchronise per step (bench.py's timed region, Model.fit between loss read-outs) enqueues several
    steps before the GPU has started the first: the optimizer's scalars (Adam's bias-corrected learning rate of step t)
    are uploaded asynchronously, so every step in flight needs a pinned row of its own. Until round 6 there was ONE: step
    t's upload read the scalars of step t + 1 .. t + 3 (scripts/step_repro.py, REPRO_SYNC=0). Here the GPU is held busy
    while eight replayed steps are enqueued; weights, moving statistics and Adam moments must equal those of eight
    synchronised steps bit for bit."""
    import yolov3
    from tf2_yolo_amd import labels
    from tf2_yolo_amd.optimizers import Adam

    def make():
        y = yolov3.Yolo((96, 96, 3), list("abcdefgh"))
        y.create_model(anchors=A9, pretrained_body=None, seed=11)
        y.model.compile(optimizer=Adam(learning_rate=1e-3), loss=y.loss())
        return y.model
    x, ys = labels.synthetic_batch(np.random.default_rng(3), 4, (96, 96), 8)
    x, ys = torch.from_numpy(x).cuda(), [torch.from_numpy(a).cuda() for a in ys]
    busy = torch.randn(4096, 4096, device="cuda")
    res = []
    for ahead in (False, True):
        m = make()
        for _ in range(3):                      # two eager steps and the recording
            m.train_step_device(x, ys)
            torch.cuda.synchronize()
        assert m._step_graphs is not None
        if ahead:
            for _ in range(60):                 # ~0.1 s of matrix products in front of the replays
                busy = (busy @ busy) * 1e-3
        for _ in range(8):
            m.train_step_device(x, ys)
            if not ahead:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        res.append((m.net.params.data.clone(), m.net.state.data.clone(), m.optimizer.m.clone(), m.optimizer.v.clone()))
    for a, b in zip(*res):
        assert torch.equal(a, b)


@pytest.mark.parametrize("ver", [1, 2, 4])
def test_training_is_bit_reproducible_for_every_version(ver):
    """Two models of one configuration from one seed, seven steps each on one batch, the host NOT waiting for the GPU between
    steps: weights, BatchNorm moving statistics and Adam moments bit-identical (scripts/step_repro.py does this at the
    benchmark's sizes). v1.5: the 7x7 RGB stem's filter gradient runs on the fp32-input kernel, whose splits met in fp32
    atomics until round 6; v2: bias in front of BatchNorm, the passthrough route; v4: Mish, SPP's overlapping pools, PAN."""
    import torch as _t
    from tf2_yolo_amd import labels
    from tf2_yolo_amd.optimizers import Adam

    def make():
        _t.manual_seed(77)
        np.random.seed(77)
        rng = np.random.default_rng(7)
        if ver == 2:
            import yolov2
            y = yolov2.Yolo((96, 96, 3), ["a", "b", "c"])
            y.create_model()
            x, ys = labels.synthetic_batch(rng, 4, (96, 96), 3, levels=1, finest_stride=32)
            loss = y.loss()
        elif ver == 1:
            import yolov1_5
            y = yolov1_5.Yolo((128, 128, 3), ["a"])
            y.create_model()
            x, ys = labels.synthetic_batch(rng, 4, (128, 128), 1, levels=1, finest_stride=64)
            loss = y.loss(binary_weight=0.5)
        else:
            import yolov4
            y = yolov4.Yolo((96, 96, 3), ["a", "b"])
            y.create_model(anchors=A9, pretrained_body=None)
            x, ys = labels.synthetic_batch(rng, 4, (96, 96), 2)
            loss = y.loss()
        y.model.compile(optimizer=Adam(learning_rate=1e-3), loss=loss)
        return y.model, _t.from_numpy(x).cuda(), [_t.from_numpy(a).cuda() for a in ys]
    res = []
    for _ in range(2):
        m, x, ys = make()
        p0 = m.net.params.data.clone()
        for _ in range(7):
            m.train_step_device(x, ys)
        _t.cuda.synchronize()
        res.append((p0, m.net.params.data.clone(), m.net.state.data.clone(), m.optimizer.m.clone(), m.optimizer.v.clone()))
    assert not _t.equal(res[0][0], res[0][1])          # (it trained)
    for a, b in zip(*res):
        assert _t.equal(a, b)


def test_batch_too_large_for_32_bit_operand_offsets_is_refused_up_front():
    """every conv kernel addresses an operand through a buffer descriptor with 32-bit offsets: a batch whose largest
    activation would reach 4 GiB is refused by Network.allocate, with the largest batch that fits, before anything is
    allocated or launched (a loud error at the boundary, not a cliff inside a kernel)"""
    import yolov3
    from tf2_yolo_amd._lib import YoloHipError
    y = yolov3.Yolo((416, 416, 3), ["a"])
    y.create_model(anchors=A9, pretrained_body=None)
    with pytest.raises(YoloHipError, match="largest per-GPU batch for this model is 192"):
        y.model.net.allocate(200)       # 416 x 416 x 32 x 4 B = 21.1 MiB per image; 193 of them (+ 1 MiB) reach 4 GiB
    assert y.model.net.batch is None
