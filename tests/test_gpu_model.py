"""End-to-end parity of the four model graphs on the GPU against the torch-CPU float64 restatement
(oracle/models.py + oracle/losses.py): inference forward, training forward (batch statistics,
moving-stat update), loss, every parameter gradient, and one Adam step.
Tolerances: forward / loss 1e-4 (north_star), or 1.5x the error an fp32 CPU run of the oracle itself has on the same
instance. Gradients: per tensor, max(2 x the fp32 CPU run's error on that tensor, 4e-4) relative to the tensor's
largest entry, against the float64 oracle evaluated with the device's LeakyReLU branch pattern (oracle/layers.py
leaky_masked): two executions of a ReLU network legitimately disagree on the sign of pre-activations
that are within rounding of zero, and one such flip moves whole gradient tensors by 1e-2; the test
asserts that the patterns differ only where |z| < 1e-4 and then compares like with like."""
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


def _reference_init(model, rng):
    """the reference's OWN initialisation of YOLOv4 (yolov4/models/backbone.py:63-111: every Conv2D
    kernel_initializer=RandomNormal(stddev=0.02), BatchNormalization defaults gamma = 1, beta = 0, moving mean 0 / variance 1;
    the bias-carrying head convs keep Keras' zeros): the case VERDICT r05 asked for beside the he-normal cases"""
    net = model.net
    p = net.params.data.cpu().numpy()
    for name in net.params.order:
        s = net.params.specs[name]
        sl = slice(s.offset, s.offset + s.size)
        if name.endswith("/kernel"):
            p[sl] = rng.standard_normal(s.size) * 0.02
        elif name.endswith("/gamma"):
            p[sl] = 1.0
        elif name.endswith("/beta") or name.endswith("/bias"):
            p[sl] = 0.0
    net.params.data.copy_(torch.from_numpy(p))
    st = net.state.data.cpu().numpy()
    for name in net.state.order:
        s = net.state.specs[name]
        st[s.offset:s.offset + s.size] = 0.0 if name.endswith("moving_mean") else 1.0
    net.state.data.copy_(torch.from_numpy(st))
    net.mark_params_changed()


def _perturb(model, rng):
    """make BN parameters / moving statistics and biases non-trivial"""
    net = model.net
    p = net.params.data.cpu().numpy()
    for name in net.params.order:
        s = net.params.specs[name]
        sl = slice(s.offset, s.offset + s.size)
        if name.endswith("/kernel"):
            # he-normal scale for every version (v4's N(0, 0.02) init makes tiny problems ill-conditioned
            # in ANY fp32 execution; the test is about arithmetic parity, not about the initialiser)
            fan_in = int(np.prod(s.shape[1:]))
            p[sl] = rng.standard_normal(s.size) * np.sqrt(2.0 / fan_in)
        elif name.endswith("/gamma"):
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


def inputs_digest(w, x, ys):
    """a short digest of an end-to-end case's inputs (weights, images, labels): the background oracle job of the headline
    tests (conftest.py) must have been given exactly what the test's own _setup produces"""
    import hashlib
    h = hashlib.sha1()
    for k in sorted(w):
        h.update(k.encode())
        h.update(np.ascontiguousarray(w[k]).tobytes()[:4096])
        h.update(np.float64(np.asarray(w[k], dtype=np.float64).sum()).tobytes())
    h.update(np.ascontiguousarray(x).tobytes()[:65536])
    for a in ys:
        h.update(np.float64(np.asarray(a, dtype=np.float64).sum()).tobytes())
    return h.hexdigest()


def log_parity_ratio(rec):
    """one line per end-to-end case: device error over the error of an fp32 CPU execution of the same oracle on the same
    inputs (forward outputs; worst parameter-gradient tensor) -> gpurun_out/parity_ratios.jsonl (committed per round as
    profiles/rNN_parity_ratios.jsonl)"""
    import json
    import os
    print("PARITY_RATIO", rec)
    try:
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "parity_ratios.jsonl"), "a") as fh:
            fh.write(json.dumps(rec) + "\n")
    except OSError:
        pass


def _weights_dict(model):
    return {f"{n}/{i}": a for n in model.layer_names() for i, a in enumerate(model.get_layer(n).get_weights())}


def _rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)


def _l2(a, b):
    a = np.asarray(a, dtype=np.float64).ravel()
    b = np.asarray(b, dtype=np.float64).ravel()
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


A6 = A9[:6]
HEADLINE_CLASSES = 80    # the full-batch headline tests (tests/test_gpu_fullsize.py, conftest.py) run the benchmark's C = 80


def _setup(version, hw=None, N=2, unbiased=True, true_c1=False, tiny=False, class_num=None, ref_init=False):
    """class_num: YOLOv3 / YOLOv4 with that many classes instead of 3 (80 = what bench.py times: 255-channel heads, the
    255 -> 256 padded head gradient, the C = 80 loss); ref_init: the reference's own YOLOv4 initialiser instead of he-normal"""
    import os
    rng = np.random.default_rng(version)
    names3 = ["a", "b", "c"] if class_num is None else [f"c{i}" for i in range(class_num)]
    C3 = len(names3)
    # v4: the 107-layer CSP/PAN chain needs >= 50 samples per BN channel at the coarsest grid to be a
    # well-conditioned fp32 problem at all (at 64x64 BOTH fp32 executions are O(1) off in the gradients)
    hw = hw or int(os.environ.get("TEST_MODEL_HW", "160" if version == 4 else "64"))
    g0 = hw // 32
    if version == 3 and tiny:
        # tiny-YOLOv3 (yolov3/models/darknet.py:107-135): two outputs (g0, 2 g0), six anchors, MaxPool(2, stride 1, same)
        import yolov3
        y = yolov3.Yolo((hw, hw, 3), ["a", "b", "c"])
        y.create_model(anchors=A6, backbone="tiny_darknet", pretrained_body=None, bn_unbiased_moving_var=unbiased)
        fwd = lambda w, x, tr, m=None: OM.yolov3_tiny_forward(w, x, A6, training=tr, leaky_masks=m,
                                                              unbiased_moving_var=unbiased)
        loss_o = [OL.wrap_yolo_loss_v3((g0 * 2 ** i, g0 * 2 ** i), 3, 3, anchors=A6[3 * i:3 * i + 3],
                                       loss_weight=[1, 1, 5, 1]) for i in range(2)]
        loss_g = y.loss()
        grids = [g0, 2 * g0]
    elif version == 3:
        import yolov3
        y = yolov3.Yolo((hw, hw, 3), names3)
        y.create_model(anchors=A9, pretrained_body=None, bn_unbiased_moving_var=unbiased)
        fwd = lambda w, x, tr, m=None: OM.yolov3_forward(w, x, A9, training=tr, leaky_masks=m, unbiased_moving_var=unbiased)
        loss_o = [OL.wrap_yolo_loss_v3((g0 * 2 ** i, g0 * 2 ** i), 3, C3, anchors=A9[3 * i:3 * i + 3],
                                       loss_weight=[1, 1, 5, 1]) for i in range(3)]
        loss_g = y.loss()
        grids = [g0, 2 * g0, 4 * g0]
    elif version == 4:
        import yolov4
        y = yolov4.Yolo((hw, hw, 3), names3)
        y.create_model(anchors=A9, pretrained_body=None, bn_unbiased_moving_var=unbiased)
        fwd = lambda w, x, tr, m=None: OM.yolov4_forward(w, x, A9, training=tr, leaky_masks=m, unbiased_moving_var=unbiased)
        loss_o = [OL.wrap_yolo_loss_v4((g0 * 2 ** i, g0 * 2 ** i), 3, C3, anchors=A9[3 * i:3 * i + 3],
                                       loss_weight=[1, 5, 1]) for i in range(3)]
        loss_g = y.loss()
        grids = [g0, 2 * g0, 4 * g0]
    elif version == 2:
        import yolov2
        y = yolov2.Yolo((hw, hw, 3), ["a", "b", "c", "d"])
        y.create_model(anchors=A5, bn_unbiased_moving_var=unbiased)
        fwd = lambda w, x, tr, m=None: OM.yolov2_forward(w, x, A5, training=tr, leaky_masks=m, unbiased_moving_var=unbiased)
        loss_o = [OL.wrap_yolo_loss_v2((g0, g0), 5, 4, A5, loss_weight=[1, 1, 5, 1])]
        loss_g = [y.loss()]
        grids = [g0]
    else:
        import yolov1_5
        # true_c1: BASELINE.json config 1 at its real size: 224x224, ONE class, B = 2, grid 4x4
        names = ["raccoon"] if true_c1 else ["a", "b"]
        y = yolov1_5.Yolo((2 * hw, 2 * hw, 3), names)
        y.create_model(bbox_num=2, bn_unbiased_moving_var=unbiased)
        if true_c1:
            g0 = y.grid_shape[0]      # 224 -> 112 -> 56 -> 28 -> 14 -> 7 -> ceil(7/2) = 4 (yolov1_5/__init__.py:91)
        assert tuple(y.grid_shape) == (g0, g0)
        fwd = lambda w, x, tr, m=None: OM.yolov1_5_forward(w, x, training=tr, leaky_masks=m, unbiased_moving_var=unbiased)
        loss_o = [OL.wrap_yolo_loss_v1((g0, g0), 2, len(names), binary_weight=0.5, loss_weight=[5, 5, 1, 1])]
        loss_g = [y.loss(binary_weight=0.5)]
        grids = [g0]
    model = y.model
    (_reference_init if ref_init else _perturb)(model, rng)
    H = y.input_shape[0]
    x = rng.random((N, H, H, 3), dtype=np.float32)
    ys = [_labels(rng, N, g, y.class_num) for g in grids]
    return y, model, fwd, loss_o, loss_g, x, ys


def _gpu_leaky_masks(net):
    """Branch pattern the device took in every LeakyReLU: sign of fma(scale, y, shift). One rounding of
    an exact product-sum keeps the exact sign, so evaluating scale*y+shift in float64 reproduces it."""
    from tf2_yolo_amd._lib import ACT_LEAKY
    masks = {}
    for u in net.units:
        if u.kind == "conv" and u.bn and u.act == ACT_LEAKY:
            scale, shift = net._bn_bufs(u)[0:2]
            z = u.y.double() * scale.double() + shift.double()
            masks[u.name] = (z > 0).cpu()
        elif u.kind == "maxpool":
            # the winner of every pooling window (flat NHWC offset into the pool's input, yolo_maxpool_fwd): the oracle takes
            # the maximum where the device took it (oracle/models.py:_Ctx.pool) -- near-ties among the 25 / 81 / 169
            # candidates of an SPP window reroute gradient entries exactly as a LeakyReLU sign flip does
            masks[u.name] = u.argmax.reshape(-1).long().cpu()
    return masks


def _kink_census(version, y_true, pred_oracle, pred_dev, class_num):
    """(object cell, anchor) pairs whose box arithmetic (oracle/losses.py:cal_iou -- tf.maximum / tf.minimum of the box
    corners, max(overlap, 0)) is within the device-oracle prediction distance of a kink: returns (#pairs near a kink, #pairs).
    A pair is near a kink when one of the four corner differences truth - prediction, or one of the two overlap extents, is
    smaller in magnitude than the largest move the straight-through substitution makes to that pair's own box."""
    yt = torch.as_tensor(y_true, dtype=torch.float64)
    g = yt.shape[1]
    if version == 1:
        B = (pred_oracle.shape[-1] - class_num) // 5
        po = pred_oracle[..., :5 * B].reshape(-1, g, g, B, 5)[..., :4]
        pd = pred_dev[..., :5 * B].reshape(-1, g, g, B, 5)[..., :4]
        t = yt[..., :4].reshape(-1, g, g, 1, 4)
        obj = yt[..., 4] > 0
    else:
        B = pred_oracle.shape[-1] // (5 + class_num)
        po = pred_oracle.reshape(-1, g, g, B, 5 + class_num)[..., :4]
        pd = pred_dev.reshape(-1, g, g, B, 5 + class_num)[..., :4]
        t = yt.reshape(-1, g, g, 1, 5 + class_num)[..., :4]
        obj = yt[..., 4] > 0
    def corners(b):
        xy = b[..., 0:2] / g
        return xy - b[..., 2:4] / 2, xy + b[..., 2:4] / 2
    tmin, tmax = corners(t)
    pmin, pmax = corners(po)
    dmin, dmax = corners(pd)
    move = torch.maximum((dmin - pmin).abs().amax(-1), (dmax - pmax).abs().amax(-1))          # [N, g, g, B]
    ov = torch.minimum(pmax, tmax) - torch.maximum(pmin, tmin)                                 # overlap extents (may be < 0)
    dist = torch.cat([(pmin - tmin).abs(), (pmax - tmax).abs(), ov.abs()], dim=-1).amin(-1)    # nearest kink
    sel = obj.unsqueeze(-1).expand_as(dist)
    return int((dist[sel] <= move[sel]).sum()), int(sel.sum())


# (version, BN moving variance fed Bessel-corrected [tf.keras fused BN, the default] or biased, C1 at its true size)
@pytest.mark.parametrize("version,unbiased,true_c1", [(3, True, False), (2, True, False), (1, True, False), (4, True, False),
                                                      (3, False, False), (1, True, True), (3, True, "tiny"),
                                                      (3, True, "416"), (4, True, "608"), (2, True, "416"),
                                                      (3, True, "tiny416"), (4, True, "608bs1"),
                                                      (3, True, "416c80bs8"), (4, True, "608refinit"),
                                                      (3, True, "416c80bs32"), (4, True, "608c80bs4")])
def test_model_parity(version, unbiased, true_c1):
    from tf2_yolo_amd import optimizers
    import conftest
    conftest.foreground_threads()     # explicit CPU threads for this test's oracle passes (half the box while conftest's jobs run)
    if true_c1 == "tiny416":   # tiny-YOLOv3 at its usual resolution: 416x416, grids 13 and 26 (bs 2)
        y, model, fwd, loss_o, loss_g, x, ys = _setup(3, hw=416, N=2, unbiased=unbiased, tiny=True)
        assert [tuple(o.shape[1:3]) for o in model.output] == [(13, 13), (26, 26)]
    elif true_c1 == "tiny":   # tiny-YOLOv3 at 96x96: grids 3 and 6, the stride-1 'same' max-pool on a 3x3 map
        y, model, fwd, loss_o, loss_g, x, ys = _setup(3, hw=96, N=4, unbiased=unbiased, tiny=True)
        assert len(model.output) == 2 and tuple(model.output[0].shape[1:3]) == (3, 3)
    elif true_c1 == "416" and version == 2:   # configs[1]: YOLOv2 Darknet-19 + passthrough at 416x416 (bs 2): 13x13 grid
        y, model, fwd, loss_o, loss_g, x, ys = _setup(2, hw=416, N=2, unbiased=unbiased)
        assert tuple(y.grid_shape) == (13, 13)
    elif true_c1 == "416":   # BASELINE.json's headline graph at its true resolution (bs 2): 13 / 26 / 52 grids, the
        # window kernels' real shapes -- with the benchmark's 80 classes since round 6 (255-channel heads, the 255 -> 256
        # zero-padded head gradient of the planes kernels, the C = 80 loss: what bench.py times; VERDICT r05 missing #3)
        y, model, fwd, loss_o, loss_g, x, ys = _setup(3, hw=416, N=2, unbiased=unbiased, class_num=80)
        assert [tuple(o.shape[1:]) for o in model.output] == [(13, 13, 255), (26, 26, 255), (52, 52, 255)]
    elif true_c1 == "416c80bs8":
        # VERDICT r05 next #2b: EVERY parameter gradient of the benchmark graph (416 x 416, C = 80) against the float64
        # oracle's autograd at batch 8 -- a quarter of the benchmark's per-GPU batch: 1.4 M-pixel planes, BatchNorm statistics
        # over 1.4 M samples, per-tensor scales from those statistics, the padded head gradients
        y, model, fwd, loss_o, loss_g, x, ys = _setup(3, hw=416, N=8, unbiased=unbiased, class_num=80)
        assert [tuple(o.shape[1:]) for o in model.output] == [(13, 13, 255), (26, 26, 255), (52, 52, 255)] and x.shape[0] == 8
    elif true_c1 == "416c80bs32":
        # ... and at the benchmark's OWN per-GPU batch of 32: the tensors bench.py times (5.5 M-pixel planes, BatchNorm
        # statistics over 5.5 M samples, per-tensor scales from them, split-K and unsplit launches as the step runs them),
        # every parameter gradient against the float64 oracle's autograd (~75 GB of host memory for the two CPU graphs)
        y, model, fwd, loss_o, loss_g, x, ys = _setup(3, hw=416, N=32, unbiased=unbiased, class_num=80)
        assert [tuple(o.shape[1:]) for o in model.output] == [(13, 13, 255), (26, 26, 255), (52, 52, 255)] and x.shape[0] == 32
    elif true_c1 == "608c80bs4":
        # YOLOv4-608 with the benchmark's 80 classes (255-channel heads, CIoU loss at C = 80) at batch 4: every parameter
        # gradient against the float64 oracle, as [3-True-416c80bs8] does for the headline graph
        y, model, fwd, loss_o, loss_g, x, ys = _setup(4, hw=608, N=4, unbiased=unbiased, class_num=80)
        assert sorted(tuple(o.shape[1:]) for o in model.output) == [(19, 19, 255), (38, 38, 255), (76, 76, 255)]
    elif true_c1 == "608refinit":
        # VERDICT r05 next #2c: YOLOv4-608 (bs 2) under the reference's OWN initialiser (N(0, 0.02) kernels, gamma 1, beta 0:
        # yolov4/models/backbone.py:63-111) instead of the he-normal kernels of the other cases; its fp32 floor is logged
        y, model, fwd, loss_o, loss_g, x, ys = _setup(4, hw=608, N=2, unbiased=unbiased, ref_init=True)
    elif true_c1 == "608bs1":
        # YOLOv4-608 at bs ONE: the case round 2 moved to bs 2 because single BN tensors were 3-5x the fp32-CPU error on
        # every conv path. Cause (scripts/grad_excess.py, profiles/r03_grad_excess_*.json): near-ties among the 25 / 81 /
        # 169 candidates of the SPP max-pool windows -- the device and the oracle picked different winners, and ONE
        # rerouted gradient entry at pan_td1_spp propagates into the whole backbone in front of it. With the oracle taking
        # the maximum where the device took it (oracle/models.py:_Ctx.pool, the max-pool twin of leaky_masked) the device
        # is at 0.9x (median) / 1.1x (90 %) / 1.8x (99 %) of the fp32-CPU error per tensor: bound 2x here.
        y, model, fwd, loss_o, loss_g, x, ys = _setup(4, hw=608, N=1, unbiased=unbiased)
    elif true_c1 == "608":   # configs[2]: YOLOv4 CSPDarknet-53 + SPP + PAN at 608x608 (bs 2): 19 / 38 / 76 grids
        y, model, fwd, loss_o, loss_g, x, ys = _setup(4, hw=608, N=2, unbiased=unbiased)
        assert sorted(tuple(o.shape[1:3]) for o in model.output) == [(19, 19), (38, 38), (76, 76)]
    elif true_c1:   # YOLOv1.5 224x224, 1 class, bs 4, B = 2: grid 4x4 -- BASELINE.json configs[0] as it is quoted
        y, model, fwd, loss_o, loss_g, x, ys = _setup(version, hw=112, N=4, unbiased=unbiased, true_c1=True)
        assert tuple(y.grid_shape) == (4, 4) and x.shape == (4, 224, 224, 3)
    else:
        y, model, fwd, loss_o, loss_g, x, ys = _setup(version, unbiased=unbiased)
    net = model.net
    w = _weights_dict(model)
    xt = torch.tensor(x, dtype=torch.float64)

    # ---- device: training forward + fused loss/grad + backward ----
    model.compile(optimizer=optimizers.Adam(learning_rate=1e-3), loss=loss_g)
    xd = torch.tensor(x).cuda()
    yd = [torch.tensor(a).cuda() for a in ys]
    outs = net.forward(xd, training=True)
    masks = _gpu_leaky_masks(net)
    dev_losses, dpred, decs = [], [], []
    for lf, o, yt in zip(loss_g, outs, yd):
        dec = torch.zeros((o.shape[0] * o.shape[1] * o.shape[2], 2), dtype=torch.int32, device="cuda")
        lo, dp = lf.fwd_bwd(yt, o, decisions=dec)      # (the kernel also reports the discrete decisions it took)
        dev_losses.append(lo[0].item())
        dpred.append(dp)
        decs.append(dec.cpu())
    net.backward(dpred)
    torch.cuda.synchronize()
    g = net.grads.cpu().numpy()

    # ---- oracle (float64), LeakyReLU branches forced to the device's pattern (layers.leaky_masked) ----
    wt = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in w.items()}
    ref_tr, ctx = fwd(wt, xt, True, masks)
    # the discrete decisions of the losses (responsible anchor = argmax IoU, ignore / truth masks) are the DEVICE's
    # (yolo_loss_fwd_bwd exports them), like the LeakyReLU branches and the max-pool winners: anchors tied at IoU 0 -- a
    # tiny intersection that one arithmetic keeps and the other rounds away -- legitimately get different winners in two
    # executions, and one different winner moves a head's gradient tensor by O(1) (oracle/losses.py:_decisions)
    dstats = {}
    ref_losses = [lf(torch.tensor(yt, dtype=torch.float64), o.detach(), decide_with=dc, stats=dstats)
                  for lf, yt, o, dc in zip(loss_o, ys, ref_tr, decs)]
    # float32 CPU execution of the same oracle (same forced branches): the error floor of ANY fp32
    # implementation on this instance (random-weight BN chains amplify rounding with depth: ~1e-4 for the
    # 72-layer v3 graph, ~1e-3 for the 107-layer v4 CSP/PAN graph; scripts/act_error_profile.py)
    w32 = {k: torch.tensor(v, requires_grad=True) for k, v in w.items()}
    out32, _ = fwd(w32, torch.tensor(x), True, masks)
    losses32 = [lf(torch.tensor(yt), o.detach(), decide_with=dc) for lf, yt, o, dc in zip(loss_o, ys, out32, decs)]
    fwd_floor = max(_rel(b32.detach().numpy(), b.detach().numpy()) for b, b32 in zip(ref_tr, out32))
    # the branch patterns may only differ where the pre-activation is within fp32 error of zero
    assert max(ctx.mask_disagree.values(), default=0.0) < max(1e-4, 10 * fwd_floor), ctx.mask_disagree
    # ... and a forced max-pool winner may only be below the true maximum by rounding
    assert max(ctx.pool_disagree.values(), default=0.0) < max(1e-4, 10 * fwd_floor), ctx.pool_disagree
    # ... and a forced loss decision only where the oracle's own IoUs are within rounding of a tie / of the threshold
    assert dstats.get("disagree", 0.0) < max(1e-4, 10 * fwd_floor), dstats
    # HOW MANY decisions were forced, per kind (the conditioning must not grow silently). Printed,
    # logged (gpurun_out/parity_census.jsonl) and bounded: a forced LeakyReLU sign needs |z| within the forward error of
    # zero (a fraction ~ 0.8 x error of unit-variance pre-activations), a forced pool winner a near-tie among <= 169
    # candidates, a forced loss decision an IoU within rounding of a tie or of a threshold
    census = {"case": f"v{version} unbiased={unbiased} {true_c1}", "fwd_floor": fwd_floor,
              "leaky_forced": sum(a for a, _ in ctx.mask_forced.values()), "leaky_total": sum(t for _, t in ctx.mask_forced.values()),
              "pool_forced": sum(a for a, _ in ctx.pool_forced.values()), "pool_total": sum(t for _, t in ctx.pool_forced.values()),
              "loss_cells": dstats.get("cells", 0), "loss_anchor_forced": dstats.get("n_anchor", 0),
              "loss_ignore_forced": dstats.get("n_ignore", 0), "loss_truth_forced": dstats.get("n_truth", 0)}
    # (measured, round 4: 1 of 1.8 M activations at 64x64, 89 of 76.6 M for YOLOv3-416, 996 of 35.3 M for YOLOv4-608; at most 9
    # pool windows; no loss decision in any case -- the bounds below are ~5x those)
    assert census["leaky_forced"] <= max(2, 0.25 * fwd_floor * census["leaky_total"]), census
    assert census["pool_forced"] <= max(2, 5e-5 * census["pool_total"]), census
    assert census["loss_anchor_forced"] + census["loss_ignore_forced"] + census["loss_truth_forced"] <= max(
        2, 1e-4 * census["loss_cells"]), census

    # The backward pass starts from the loss gradient AT THE DEVICE'S predictions (straight-through: value of the device,
    # graph of the oracle). The loss has more kinks than its argmax -- max / min of box corners, max(overlap, 0), the clips
    # -- and an object whose predicted box touches its ground truth within 1e-4 of zero overlap puts the oracle's own
    # predictions (1e-4 from the device's) on the other side of one: O(1) in that cell's xy gradient, 1e-2 in a head tensor
    # (seen on out3_box1_xy_conv of YOLOv4-608, scripts/loss_cell_debug.py: at the device's predictions the kernel's
    # gradient equals the oracle's to 1e-6). The forward outputs and loss VALUES are compared at the oracle's own predictions.
    # Guard (VERDICT r03 weak #1d): the substitution may move a prediction only by the forward error -- asserted BEFORE
    # substituting -- and the census records how many (object cell, anchor) pairs sit within that distance of a kink of
    # the box arithmetic (corner ties of min / max, zero overlap), i.e. how many cells the substitution can matter for.
    dev_pred = [o.detach().double().cpu() for o in outs]
    st_dist = 0.0
    for dp_, o in zip(dev_pred, ref_tr):
        st_dist = max(st_dist, float((dp_ - o.detach()).abs().max()) / max(float(o.detach().abs().max()), 1e-30))
    assert st_dist < max(1e-4, 1.5 * fwd_floor), (st_dist, fwd_floor)
    census["straight_through_max_rel_move"] = st_dist
    near, pairs = 0, 0
    for yt, o, dp_ in zip(ys, ref_tr, dev_pred):
        k_, p_ = _kink_census(version, yt, o.detach(), dp_, len(y.class_names))
        near, pairs = near + k_, pairs + p_
    census["object_anchor_pairs"], census["pairs_within_move_of_a_box_kink"] = pairs, near
    sum(lf(torch.tensor(yt, dtype=torch.float64), o + (dp_ - o).detach(), decide_with=dc)
        for lf, yt, o, dc, dp_ in zip(loss_o, ys, ref_tr, decs, dev_pred)).backward()
    sum(lf(torch.tensor(yt), o + (dp_.float() - o).detach(), decide_with=dc)
        for lf, yt, o, dc, dp_ in zip(loss_o, ys, out32, decs, dev_pred)).backward()

    # forward outputs / losses: 1e-4 (north_star), or 1.5x the error of the fp32 CPU execution on this instance (measured
    # worst device / fp32-CPU ratio over all cases: 1.16 -- profiles/r05_parity_ratios.jsonl; the bar was 3x until round 4)
    fwd_err = max(_rel(a.cpu().numpy(), b.detach().numpy()) for a, b in zip(outs, ref_tr))
    assert fwd_err < max(1e-4, 1.5 * fwd_floor), (fwd_err, fwd_floor)
    for dl, rl, l32 in zip(dev_losses, ref_losses, losses32):
        tol = max(1e-4, 1.5 * abs(l32.item() - rl.item()) / max(abs(rl.item()), 1.0))
        assert abs(dl - rl.item()) < tol * max(abs(rl.item()), 1.0)

    # moving statistics after the training forward
    for bn_name, (mm, mv) in ctx.moving.items():
        got = model.get_layer(bn_name).get_weights()
        assert _rel(got[2], mm.detach().numpy()) < 1e-4 and _rel(got[3], mv.detach().numpy()) < 1e-4

    # every parameter gradient, per tensor, relative to the tensor's largest entry
    gmax = max(float(t.grad.abs().max()) for t in wt.values() if t.grad is not None)
    worst = ("", 0.0, 0.0)
    worst_ratio = ("", 0.0, 0.0, 0.0)
    for n in model.layer_names():
        layer_w = model.get_layer(n).get_weights()
        if not layer_w or n.endswith("_anchor"):
            continue
        for i in range(len(layer_w)):
            r = wt[f"{n}/{i}"].grad
            if r is None:
                continue   # moving statistics
            got = _grad_view(model, n, i, g)
            if n.endswith("_conv") and i == 1 and f"{n[:-5]}_bn/0" in w:
                # conv bias in front of BatchNormalization (v1.5 / v2): the true gradient is exactly 0
                assert np.abs(got).max() < 1e-4 * gmax, (n, np.abs(got).max())
                continue
            e = _rel(got, r.numpy())
            e32 = _rel(w32[f"{n}/{i}"].grad.numpy(), r.numpy())
            worst = max(worst, (n, e, e32), key=lambda t: t[1])
            worst_ratio = max(worst_ratio, (f"{n}/{i}", e / max(e32, 2e-4), e, e32), key=lambda t: t[1])
            # EVERY case (round 5; until round 4 only YOLOv4-608 bs 1 had such a bound, the others max(1e-3, 4 e32, 3 floor)): at
            # most twice the error of the fp32 CPU execution of the oracle on the same tensor, with an absolute floor of 4e-4
            # of the tensor's largest entry (a handful of head biases sit at 1e-4 where the CPU happens to be at 2e-5; measured
            # worst: YOLOv4-608 bs 2, out1_box1_prob_conv bias, 3.1e-4 against the CPU's 1.25e-4 -- profiles/r05_parity_ratios.jsonl)
            assert e < max(2 * e32, 4e-4), (n, i, e, e32)
    print("worst gradient error", worst)
    log_parity_ratio({"case": f"v{version} unbiased={unbiased} {true_c1}", "classes": len(y.class_names), "batch": int(x.shape[0]),
                      "fp32_floor": fwd_floor, "forward_err": fwd_err,
                      "forward_meets_plain_1e-4": bool(fwd_err < 1e-4), "fp32_cpu_meets_plain_1e-4": bool(fwd_floor < 1e-4),
                      "forward_ratio": fwd_err / max(fwd_floor, 1e-30), "worst_gradient_tensor": worst_ratio[0],
                      "worst_gradient_err": worst_ratio[2], "fp32_cpu_err_same_tensor": worst_ratio[3],
                      "worst_gradient_ratio_err_over_max_e32_2e-4": worst_ratio[1],
                      "largest_gradient_err": worst[1], "largest_gradient_err_fp32_cpu": worst[2]})

    # ---- unconditioned companion (VERDICT r03 next #2b): the oracle makes ALL its own decisions -- its own LeakyReLU signs,
    # pool winners, responsible anchors and masks, its own predictions -- and the device's gradients are compared with it per
    # tensor in the L2 norm (one rerouted entry moves max-norm errors by 1e-2 but not the L2 distance of a whole tensor). The
    # fraction of entries beyond 1e-4 of the tensor's maximum goes to the log, so drift of the conditioned test's forcing
    # would show here. Run on four cases -- three where nothing is forced (the assertion applies) and YOLOv3 at 64x64 (one
    # forced sign); the two extra CPU passes cost as much as the conditioned test itself and the GPU suite has to stay within
    # ten minutes (the figures of the larger cases, YOLOv3-416 among them: profiles/r04_b_parity_census.jsonl).
    if (version, unbiased, true_c1) in ((2, True, False), (1, True, False), (3, True, "tiny"), (3, True, False)):
        wf = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in w.items()}
        free_tr, _ = fwd(wf, xt, True, None)
        sum(lf(torch.tensor(yt, dtype=torch.float64), o) for lf, yt, o in zip(loss_o, ys, free_tr)).backward()
        # ... and beside it the SAME unconditioned comparison for an fp32 CPU execution of the oracle (logged). What can be
        # asserted: where NOTHING was forced -- the device took every discrete decision exactly as the float64 oracle does --
        # the unconditioned comparison is the conditioned one and must hold 1e-3 in L2. Where a decision was forced the
        # distance measures the flip, not the arithmetic: ONE forced sign among 1.8 M activations (YOLOv3 at 64x64, 2 images)
        # puts conv1's kernel gradient 2.8e-2 away in L2 while the conditioned error of the same tensor is 4e-4 -- measured
        # round 4, which is the reason the conditioned test exists; those cases log both distances and the flip counts.
        wf32 = {k: torch.tensor(v, requires_grad=True) for k, v in w.items()}
        free32, _ = fwd(wf32, torch.tensor(x), True, None)
        sum(lf(torch.tensor(yt), o) for lf, yt, o in zip(loss_o, ys, free32)).backward()
        worst_l2, worst_cpu, beyond, beyond_cpu, total = ("", 0.0, 0.0), 0.0, 0, 0, 0
        for n in model.layer_names():
            layer_w = model.get_layer(n).get_weights()
            if not layer_w or n.endswith("_anchor"):
                continue
            for i in range(len(layer_w)):
                r = wf[f"{n}/{i}"].grad
                if r is None or (n.endswith("_conv") and i == 1 and f"{n[:-5]}_bn/0" in w):
                    continue
                got, ref_g, cpu_g = _grad_view(model, n, i, g), r.numpy(), wf32[f"{n}/{i}"].grad.numpy()
                e_dev, e_cpu = _l2(got, ref_g), _l2(cpu_g, ref_g)
                worst_l2 = max(worst_l2, (f"{n}/{i}", e_dev, e_cpu), key=lambda t: t[1])
                worst_cpu = max(worst_cpu, e_cpu)
                beyond += int((np.abs(got - ref_g) > 1e-4 * np.abs(ref_g).max()).sum())
                beyond_cpu += int((np.abs(cpu_g - ref_g) > 1e-4 * np.abs(ref_g).max()).sum())
                total += ref_g.size
        census["unconditioned_worst_l2_rel"], census["unconditioned_worst_tensor"] = worst_l2[1], worst_l2[0]
        census["unconditioned_worst_l2_rel_fp32_cpu"] = worst_cpu
        census["unconditioned_frac_entries_beyond_1e-4_of_max"] = beyond / max(total, 1)
        census["unconditioned_frac_entries_beyond_1e-4_of_max_fp32_cpu"] = beyond_cpu / max(total, 1)
        n_forced = (census["leaky_forced"] + census["pool_forced"] + census["loss_anchor_forced"] + census["loss_ignore_forced"]
                    + census["loss_truth_forced"])
        if n_forced == 0:
            assert worst_l2[1] < 1e-3, worst_l2
        print("unconditioned oracle:", n_forced, "forced decisions; worst per-tensor L2-relative gradient error (tensor, device, fp32 CPU on that tensor)", worst_l2,
              "worst fp32-CPU tensor", worst_cpu, "| entries beyond 1e-4 of their tensor's maximum: device", beyond, "fp32 CPU",
              beyond_cpu, "of", total)
    print("PARITY_CENSUS", census)
    try:
        import json
        import os
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        os.makedirs(os.path.join(root, "gpurun_out"), exist_ok=True)
        with open(os.path.join(root, "gpurun_out", "parity_census.jsonl"), "a") as fh:
            fh.write(json.dumps(census) + "\n")
    except OSError:
        pass

    # ---- inference forward: moving statistics := this batch's statistics (keeps the net well scaled) ----
    w_inf = dict(w)
    for bn_name, (mm, mv) in ctx.moving.items():
        mean_b = (mm.detach().numpy() - 0.99 * w[f"{bn_name}/2"]) / 0.01
        var_b = (mv.detach().numpy() - 0.99 * w[f"{bn_name}/3"]) / 0.01
        gam, bet = w[f"{bn_name}/0"], w[f"{bn_name}/1"]
        model.get_layer(bn_name).set_weights([gam, bet, mean_b, var_b])
        w_inf[f"{bn_name}/2"], w_inf[f"{bn_name}/3"] = mean_b.astype(np.float32), var_b.astype(np.float32)
    pred = model.predict(x)
    pred = pred if isinstance(pred, list) else [pred]
    ref, _ = fwd(w_inf, xt, False)
    ref32, _ = fwd({k: torch.tensor(v, dtype=torch.float32) for k, v in w_inf.items()}, torch.tensor(x), False)
    inf_floor = max(_rel(b32.numpy(), b.numpy()) for b, b32 in zip(ref, ref32))
    inf_err = max(_rel(a, b.numpy()) for a, b in zip(pred, ref))
    log_parity_ratio({"case": f"v{version} unbiased={unbiased} {true_c1} (inference forward)", "fp32_floor": inf_floor,
                      "forward_err": inf_err, "forward_ratio": inf_err / max(inf_floor, 1e-30)})
    for a, b in zip(pred, ref):
        assert a.shape == tuple(b.shape)
        assert np.isfinite(a).all()
        # 1e-4, or 1.5x the error of an fp32 CPU execution of the oracle on this small-batch problem
        assert _rel(a, b.numpy()) < max(1e-4, 1.5 * inf_floor)

    # ---- one Adam step ----
    net.forward(xd, training=True)       # restore training state consumed by predict()
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


def test_planes_path_agrees_with_exact_kernels_at_full_layer_sizes():
    """YOLOv3-416 (the benchmark's layer sizes, batch 4): the fp16 x 3 planes path against the exact bf16 x 6
    kernels (YOLO_CONV_PLANES=0), each in its own process. The unit tests run small tensors; this one exercises
    the per-tensor scales and bounds on 5.5 M-pixel activations. Head outputs and losses must agree to 1e-4
    (fp32 parity bar); gradient norms to 1e-3 (LeakyReLU branch flips near zero move individual gradients)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = []
    for planes in ("1", "0"):
        env = dict(os.environ, YOLO_CONV_PLANES=planes)
        out = subprocess.run([sys.executable, os.path.join(root, "scripts", "full_size_step.py"), "4"], env=env,
                             capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        res.append(json.loads(out.stdout.strip().splitlines()[-1]))
    a, b = res
    assert a["grad_finite"] and b["grad_finite"]
    for la, lb in zip(a["loss"], b["loss"]):
        assert abs(la - lb) <= 1e-4 * abs(lb), (a["loss"], b["loss"])
    for oa, ob in zip(a["out_abs_sum"], b["out_abs_sum"]):
        assert abs(oa - ob) <= 1e-4 * abs(ob), (a["out_abs_sum"], b["out_abs_sum"])
    assert abs(a["grad_l2"] - b["grad_l2"]) <= 1e-3 * b["grad_l2"], (a["grad_l2"], b["grad_l2"])


@pytest.mark.parametrize("version", [3, 4])
def test_fused_bn_backward_reduction_option_gives_the_same_gradients(version, monkeypatch):
    """YOLO_BN_FUSED_REDUCE=1 (the BatchNormalization-backward sums made by the data gradient that completes dL/d(a):
    engine._plan_fused_reduce, include/yolo_hip.h: yolo_conv2d_dgrad_planes_bnred) against the default (standalone
    reduction): same weights, same batch, one forward + loss + backward. Every parameter gradient within 2e-5 of the
    largest gradient of its tensor (fp32 tile sums folded in fp64 against per-element fp64 accumulation; split-K data
    gradients run unsplit in the fused form), the losses equal, and the fused form actually in use."""
    import numpy as np
    from tf2_yolo_amd import graphs, labels
    import yolov3, yolov4
    hw, cls = 96, 5
    rng = np.random.default_rng(11)
    x_h, ys_h = labels.synthetic_batch(rng, 3, (hw, hw), cls)
    x = torch.from_numpy(x_h).cuda()
    ys = [torch.from_numpy(y).cuda() for y in ys_h]
    res = []
    for mode in ("0", "1"):
        monkeypatch.setenv("YOLO_BN_FUSED_REDUCE", mode)
        if version == 3:
            y = yolov3.Yolo((hw, hw, 3), [f"c{i}" for i in range(cls)])
            y.create_model(pretrained_body=None, seed=5)
        else:
            y = yolov4.Yolo((hw, hw, 3), [f"c{i}" for i in range(cls)])
            y.create_model(anchors=graphs.V4_DEFAULT_ANCHORS, pretrained_body=None, seed=5)
        m = y.model
        lossf = y.loss()
        net = m.net
        outs = net.forward(x, training=True)
        dpred = [torch.empty_like(o) for o in outs]
        lb = [torch.zeros(8, device="cuda", dtype=torch.float64) for _ in outs]
        for i, (o, yt) in enumerate(zip(outs, ys)):
            lossf[i].fwd_bwd(yt, o, grad_scale=1.0, dpred=dpred[i], loss_out=lb[i])
        net.grads.zero_()
        net.backward(dpred)
        torch.cuda.synchronize()
        fused_units = sum(1 for u in net.units if getattr(u, "bnred", None) is not None)
        res.append((net.grads.clone(), [float(b[0]) for b in lb], fused_units, net))
    (g0, l0, n0, net0), (g1, l1, n1, _) = res
    assert n0 == 0 and n1 >= 20
    assert l0 == pytest.approx(l1, rel=1e-12)   # (the forward pass is the same code in both: fp64 sums, atomic order aside)
    for name in net0.params.order:
        s = net0.params.specs[name]
        a, b = g0[s.offset:s.offset + s.size], g1[s.offset:s.offset + s.size]
        scale = max(a.abs().max().item(), 1e-30)
        assert (a - b).abs().max().item() / scale < 2e-5, name


def test_dy_planes_per_layer_or_shared_give_identical_gradients(monkeypatch):
    """YOLO_DYP_PER_LAYER=1 (default since round 5: every conv layer owns the planes of its d(conv out), no event wait on the
    filter-gradient stream per layer) against the two shared, alternately used buffers of rounds 1-4: the same kernels on the
    same values in the same order -- every parameter gradient bit-identical, over two steps (the second step re-uses the
    buffers the first one's filter gradients read)."""
    import numpy as np
    from tf2_yolo_amd import labels
    import yolov3
    hw, cls = 96, 4
    x_h, ys_h = labels.synthetic_batch(np.random.default_rng(21), 2, (hw, hw), cls)
    x = torch.from_numpy(x_h).cuda()
    ys = [torch.from_numpy(y).cuda() for y in ys_h]
    grads = []
    for mode in ("0", "1"):
        monkeypatch.setenv("YOLO_DYP_PER_LAYER", mode)
        y = yolov3.Yolo((hw, hw, 3), [f"c{i}" for i in range(cls)])
        y.create_model(pretrained_body=None, seed=9)
        net, lossf = y.model.net, y.loss()
        assert bool(net._dyp_per_layer) == (mode == "1")
        for _ in range(2):
            outs = net.forward(x, training=True)
            dpred = [torch.empty_like(o) for o in outs]
            for i, (o, yt) in enumerate(zip(outs, ys)):
                lossf[i].fwd_bwd(yt, o, grad_scale=1.0, dpred=dpred[i])
            net.grads.zero_()
            net.backward(dpred)
        torch.cuda.synchronize()
        assert (len(net._dyp_own) > 50) == (mode == "1")
        grads.append(net.grads.clone())
    assert torch.equal(grads[0], grads[1])


@pytest.mark.parametrize("version", [3, 4])
def test_bn_fold_option_gives_the_same_gradients(version, monkeypatch):
    """YOLO_BN_FOLD=1 (the BatchNormalization-backward reduction finished by its own launch: include/yolo_hip.h,
    yolo_bn_act_bwd_reduce_fold_ld) against the default two-launch form: same weights, same batch, one forward + loss +
    backward, twice. Every parameter gradient within 1e-6 of the largest gradient of its tensor (the two forms add the same
    fp64 partial sums in different fixed orders), and the folded form is bit-identical from run to run."""
    import numpy as np
    from tf2_yolo_amd import graphs, labels
    import yolov3, yolov4
    hw, cls = 96, 5
    x_h, ys_h = labels.synthetic_batch(np.random.default_rng(12), 3, (hw, hw), cls)
    x = torch.from_numpy(x_h).cuda()
    ys = [torch.from_numpy(y).cuda() for y in ys_h]
    res = []
    for mode in ("0", "1", "1"):
        monkeypatch.setenv("YOLO_BN_FOLD", mode)
        if version == 3:
            y = yolov3.Yolo((hw, hw, 3), [f"c{i}" for i in range(cls)])
            y.create_model(pretrained_body=None, seed=5)
        else:
            y = yolov4.Yolo((hw, hw, 3), [f"c{i}" for i in range(cls)])
            y.create_model(anchors=graphs.V4_DEFAULT_ANCHORS, pretrained_body=None, seed=5)
        net, lossf = y.model.net, y.loss()
        assert bool(net._bn_fold) == (mode == "1")
        for _ in range(2):      # the second pass re-uses the ticket words the first one left at zero
            outs = net.forward(x, training=True)
            dpred = [torch.empty_like(o) for o in outs]
            for i, (o, yt) in enumerate(zip(outs, ys)):
                lossf[i].fwd_bwd(yt, o, grad_scale=1.0, dpred=dpred[i])
            net.grads.zero_()
            net.backward(dpred)
        torch.cuda.synchronize()
        res.append((net.grads.clone(), net))
    (g0, net0), (g1, _), (g2, _) = res
    assert torch.equal(g1, g2)
    for name in net0.params.order:
        s = net0.params.specs[name]
        a, b = g0[s.offset:s.offset + s.size], g1[s.offset:s.offset + s.size]
        assert (a - b).abs().max().item() / max(a.abs().max().item(), 1e-30) < 1e-6, name
