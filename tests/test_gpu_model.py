"""End-to-end parity of the four model graphs on the GPU against the torch-CPU float64 restatement
(oracle/models.py + oracle/losses.py): inference forward, training forward (batch statistics,
moving-stat update), loss, every parameter gradient, and one Adam step.
Tolerances: forward / loss 1e-4 (north_star); gradients are compared per tensor relative to that
tensor's largest entry with 2e-3 (fp32 accumulation through up to 107 BN layers at 2x2 grids)."""
import numpy as np
import pytest
import torch

from oracle import losses as OL
from oracle import models as OM

pytestmark = pytest.mark.gpu

A9 = [[0.89663461, 0.78365384], [0.375, 0.47596153], [0.27884615, 0.21634615], [0.14182692, 0.28605769],
      [0.14903846, 0.10817307], [0.07211538, 0.14663461], [0.07932692, 0.05528846], [0.03846153, 0.07211538],
      [0.02403846, 0.03125]]
A5 = [[0.75157846, 0.70525231], [0.60637077, 0.27136769], [0.25680231, 0.42110308], [0.14418923, 0.15865615],
      [0.04405615, 0.05210654]]


def _labels(rng, N, g, C):
    yt = np.zeros((N, g, g, 5 + C), dtype=np.float32)
    mask = rng.random((N, g, g)) < 0.3
    mask[0, 0, 0] = True
    n = int(mask.sum())
    yt[mask, 0:2] = rng.random((n, 2))
    yt[mask, 2:4] = rng.random((n, 2)) * 0.5 + 0.05
    yt[mask, 4] = 1
    oh = np.zeros((n, C), dtype=np.float32)
    oh[np.arange(n), rng.integers(0, C, n)] = 1
    yt[mask, 5:] = oh
    return yt


def _perturb(model, rng):
    """make BN parameters / moving statistics and biases non-trivial"""
    net = model.net
    p = net.params.data.cpu().numpy()
    for name in net.params.order:
        s = net.params.specs[name]
        sl = slice(s.offset, s.offset + s.size)
        if name.endswith("/gamma"):
            p[sl] = 1 + 0.2 * rng.standard_normal(s.size)
        elif name.endswith("/beta") or name.endswith("/bias"):
            p[sl] = 0.1 * rng.standard_normal(s.size)
    net.params.data.copy_(torch.from_numpy(p))
    st = net.state.data.cpu().numpy()
    for name in net.state.order:
        s = net.state.specs[name]
        sl = slice(s.offset, s.offset + s.size)
        st[sl] = (0.1 * rng.standard_normal(s.size)) if name.endswith("moving_mean") else (0.5 + rng.random(s.size))
    net.state.data.copy_(torch.from_numpy(st))
    net.mark_params_changed()


def _weights_dict(model):
    return {f"{n}/{i}": a for n in model.layer_names() for i, a in enumerate(model.get_layer(n).get_weights())}


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _setup(version):
    rng = np.random.default_rng(version)
    N = 2
    if version == 3:
        import yolov3
        y = yolov3.Yolo((64, 64, 3), ["a", "b", "c"])
        y.create_model(anchors=A9, pretrained_body=None)
        fwd = lambda w, x, tr: OM.yolov3_forward(w, x, A9, training=tr)
        loss_o = [OL.wrap_yolo_loss_v3((2 * 2 ** i, 2 * 2 ** i), 3, 3, anchors=A9[3 * i:3 * i + 3],
                                       loss_weight=[1, 1, 5, 1]) for i in range(3)]
        loss_g = y.loss()
        grids = [2, 4, 8]
    elif version == 4:
        import yolov4
        y = yolov4.Yolo((64, 64, 3), ["a", "b", "c"])
        y.create_model(anchors=A9, pretrained_body=None)
        fwd = lambda w, x, tr: OM.yolov4_forward(w, x, A9, training=tr)
        loss_o = [OL.wrap_yolo_loss_v4((2 * 2 ** i, 2 * 2 ** i), 3, 3, anchors=A9[3 * i:3 * i + 3],
                                       loss_weight=[1, 5, 1]) for i in range(3)]
        loss_g = y.loss()
        grids = [2, 4, 8]
    elif version == 2:
        import yolov2
        y = yolov2.Yolo((64, 64, 3), ["a", "b", "c", "d"])
        y.create_model(anchors=A5)
        fwd = lambda w, x, tr: OM.yolov2_forward(w, x, A5, training=tr)
        loss_o = [OL.wrap_yolo_loss_v2((2, 2), 5, 4, A5, loss_weight=[1, 1, 5, 1])]
        loss_g = [y.loss()]
        grids = [2]
    else:
        import yolov1_5
        y = yolov1_5.Yolo((128, 128, 3), ["a", "b"])
        y.create_model(bbox_num=2)
        assert tuple(y.grid_shape) == (2, 2)
        fwd = lambda w, x, tr: OM.yolov1_5_forward(w, x, training=tr)
        loss_o = [OL.wrap_yolo_loss_v1((2, 2), 2, 2, binary_weight=0.5, loss_weight=[5, 5, 1, 1])]
        loss_g = [y.loss(binary_weight=0.5)]
        grids = [2]
    model = y.model
    _perturb(model, rng)
    H = y.input_shape[0]
    x = rng.random((N, H, H, 3), dtype=np.float32)
    ys = [_labels(rng, N, g, y.class_num) for g in grids]
    return y, model, fwd, loss_o, loss_g, x, ys


@pytest.mark.parametrize("version", [3, 2, 1, 4])
def test_model_parity(version):
    from tf2_yolo_amd import optimizers
    y, model, fwd, loss_o, loss_g, x, ys = _setup(version)
    net = model.net
    w = _weights_dict(model)
    xt = torch.tensor(x, dtype=torch.float64)

    # ---- inference forward (moving statistics) ----
    pred = model.predict(x)
    pred = pred if isinstance(pred, list) else [pred]
    ref, _ = fwd(w, xt, False)
    for a, b in zip(pred, ref):
        assert a.shape == tuple(b.shape)
        assert _rel(a, b.numpy()) < 1e-4

    # ---- training forward + loss + backward ----
    wt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in w.items()}
    ref_tr, moving = fwd(wt, xt, True)
    ref_losses = [lf(torch.tensor(yt, dtype=torch.float64), o) for lf, yt, o in zip(loss_o, ys, ref_tr)]
    total = sum(ref_losses)
    total.backward()

    model.compile(optimizer=optimizers.Adam(learning_rate=1e-3), loss=loss_g)
    xd = torch.tensor(x).cuda()
    yd = [torch.tensor(a).cuda() for a in ys]
    outs = net.forward(xd, training=True)
    for a, b in zip(outs, ref_tr):
        assert _rel(a.cpu().numpy(), b.detach().numpy()) < 1e-4
    dpred = []
    for lf, o, yt, rl in zip(loss_g, outs, yd, ref_losses):
        lo, dp = lf.fwd_bwd(yt, o)
        assert abs(lo[0].item() - rl.item()) < 1e-4 * max(abs(rl.item()), 1.0)
        dpred.append(dp)
    net.backward(dpred)
    torch.cuda.synchronize()

    # moving statistics after the training forward
    for bn_name, (mm, mv) in moving.items():
        got = model.get_layer(bn_name).get_weights()
        assert _rel(got[2], mm.numpy()) < 1e-4 and _rel(got[3], mv.numpy()) < 1e-4

    # every parameter gradient
    g = net.grads.cpu().numpy()
    worst = 0.0
    for n in model.layer_names():
        layer_w = model.get_layer(n).get_weights()
        if not layer_w or n.endswith("_anchor"):
            continue
        # read the gradient through the same Keras-layout views by swapping buffers
        refs = [wt[f"{n}/{i}"].grad for i in range(len(layer_w))]
        for i, r in enumerate(refs):
            if r is None:
                continue   # moving statistics
            got = _grad_view(model, n, i, g)
            e = _rel(got, r.numpy())
            worst = max(worst, e)
            assert e < 2e-3, (n, i, e)

    # ---- one Adam step ----
    p_before = net.params.data.clone()
    model.optimizer.step()
    torch.cuda.synchronize()
    assert float(net.grads.abs().max()) == 0.0
    lr, b1, b2, eps = 1e-3, 0.9, 0.999, 1e-7
    gt = torch.from_numpy(g).cuda().double()
    m = (1 - b1) * gt
    v = (1 - b2) * gt * gt
    lr_t = lr * np.sqrt(1 - b2) / (1 - b1)
    expect = p_before.double() - lr_t * m / (v.sqrt() + eps)
    assert (net.params.data.double() - expect).abs().max().item() < 1e-6


def _grad_view(model, lname, idx, flat):
    """gradient of Keras weight `idx` of layer `lname`, in Keras layout, from the flat grad buffer"""
    net = model.net
    for u in net.units:
        if u.kind == "conv" and lname == f"{u.name}_conv":
            if idx == 0:
                s = u.p_kernel
                return np.transpose(flat[s.offset:s.offset + s.size].reshape(s.shape), (1, 2, 3, 0))
            s = u.p_bias
            return flat[s.offset:s.offset + s.size]
        if u.kind == "conv" and u.bn and lname == f"{u.name}_bn":
            s = (u.p_gamma, u.p_beta)[idx]
            return flat[s.offset:s.offset + s.size]
        if u.kind == "head" and lname in model._head_layer_names(u):
            lo, hi = model._head_rows(u, lname)
            cin = u.src.c
            if idx == 0:
                s = u.p_kernel
                return flat[s.offset:s.offset + s.size].reshape(-1, cin)[lo:hi].T.reshape(1, 1, cin, hi - lo)
            s = u.p_bias
            return flat[s.offset:s.offset + s.size][lo:hi]
    raise KeyError(lname)


def test_param_counts_match_published_architecture():
    """YOLOv3-416 C=80 has 62 001 757 parameters (61 949 149 trainable), the
    keras-yolo3 figure; pins the graph reading independently of any restatement."""
    import yolov3
    y = yolov3.Yolo((416, 416, 3), ["c"] * 80)
    y.create_model(pretrained_body=None)
    assert y.model.trainable_count() == 61949149
    assert y.model.count_params() == 62001757
    assert [tuple(o.shape[1:]) for o in y.model.output] == [(13, 13, 255), (26, 26, 255), (52, 52, 255)]


def test_weights_roundtrip_and_layer_views(tmp_path):
    import yolov3
    y = yolov3.Yolo((64, 64, 3), ["a", "b"])
    y.create_model(anchors=A9, pretrained_body=None)
    m = y.model
    k, = m.get_layer("conv1_conv").get_weights()
    assert k.shape == (3, 3, 3, 32)
    kxy, bxy = m.get_layer("out2_box3_xy_conv").get_weights()
    assert kxy.shape == (1, 1, 512, 2) and bxy.shape == (2,)
    m.get_layer("out2_box3_xy_conv").set_weights([kxy + 1, bxy + 2])
    k2, b2 = m.get_layer("out2_box3_xy_conv").get_weights()
    assert np.allclose(k2, kxy + 1) and np.allclose(b2, bxy + 2)
    path = str(tmp_path / "w.npz")
    m.save_weights(path)
    y2 = yolov3.Yolo((64, 64, 3), ["a", "b"])
    y2.create_model(anchors=A9, pretrained_body=None, seed=99)
    y2.model.load_weights(path)
    x = np.random.default_rng(0).random((1, 64, 64, 3), dtype=np.float32)
    for a, b in zip(m.predict(x), y2.model.predict(x)):
        assert np.array_equal(a, b)
    with pytest.raises(ValueError, match="Invalid backbone"):
        yolov3.Yolo((64, 64, 3), ["a"]).create_model(backbone="nope", pretrained_body=None)
    with pytest.raises(ValueError, match="multiple"):
        yolov3.Yolo((64, 64, 3), ["a"]).create_model(anchors=A9[:8], pretrained_body=None)
