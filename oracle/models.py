"""Model graphs restated on torch CPU (float64 by default) from the reference definitions:

  v3   yolov3/models/backbone.py:27-95, yolov3/models/darknet.py:71-104, yolov3/models/__init__.py:13-70
  v4   yolov4/models/backbone.py:22-185, yolov4/models/darknet.py:72-146, yolov4/models/__init__.py:14-71
  v2   yolov2/models/backbone.py:11-73, yolov2/models/darknet.py:32-106
  v1.5 yolov1_5/models/backbone.py:9-48, yolov1_5/models/darknet.py:26-55

Weights come in as {"<keras layer name>/<index>": ndarray} in Keras layouts (Conv kernel HWIO
[, bias]; BN gamma, beta, moving_mean, moving_variance), i.e. what Model.save_weights writes.
v1.5 / v2 layers are unnamed in the reference; the names used here are this repo's.
Every function returns (outputs, ctx); ctx.moving maps bn layer name -> (moving_mean,
moving_variance) after the Keras update when training=True.
"""
import torch

from . import layers as L


KEEP_ACTS = True    # module switch read by every forward: False = do not retain per-unit activations (full-batch runs)


class _Ctx:
    def __init__(self, weights, training, dtype, unbiased_moving_var=False, leaky_masks=None):
        self.w = {k: torch.as_tensor(v, dtype=dtype) if not torch.is_tensor(v) else v for k, v in weights.items()}
        self.training = training
        self.moving = {}
        self.unbiased = unbiased_moving_var
        self.leaky_masks = leaky_masks      # {unit name: bool tensor}: see layers.leaky_masked
        self.mask_disagree = {}             # unit name -> largest |z| where the forced branch differs
        self.pool_disagree = {}             # max-pool unit name -> largest (true max - forced winner)
        self.mask_forced = {}               # unit name -> (activations whose branch was forced against this execution's sign, all)
        self.pool_forced = {}               # max-pool unit name -> (windows whose forced winner is not this execution's maximum, all)
        self.acts = {}                      # unit name -> activation (debug / per-layer parity)
        self.keep_acts = KEEP_ACTS          # False: memory-light runs at full batch (tests/test_gpu_fullsize.py)

    def conv(self, x, name, stride=1, padding="same", bias=False):
        b = self.w[f"{name}/1"] if bias else None
        return L.conv2d(x, self.w[f"{name}/0"], b, stride=stride, padding=padding)

    def bn(self, x, name):
        gamma, beta = self.w[f"{name}/0"], self.w[f"{name}/1"]
        mm, mv = self.w[f"{name}/2"], self.w[f"{name}/3"]
        if self.training:
            y, mean, var = L.batchnorm_train(x, gamma, beta)
            n = x.numel() // x.shape[-1]
            fed = var * n / (n - 1) if self.unbiased else var
            self.moving[name] = (L.moving_update(mm, mean.detach()), L.moving_update(mv, fed.detach()))
            return y
        return L.batchnorm_infer(x, gamma, beta, mm, mv)

    def pool(self, x, name, k, stride, padding="valid"):
        """MaxPooling2D. With a device argmax pattern under `name` in leaky_masks (flat NHWC offsets into x, what
        yolo_maxpool_fwd records) the maximum is taken WHERE THE DEVICE TOOK IT -- the max-pool twin of leaky_masked:
        two executions legitimately pick different winners among values within rounding of each other (a 13x13 SPP
        window holds 169 candidates), and one different winner reroutes a gradient entry. pool_disagree[name] = the
        largest amount by which a forced winner is below the true maximum (the caller asserts it is rounding-sized)."""
        y = L.maxpool(x, k, stride, padding)
        if self.leaky_masks is None or name not in self.leaky_masks:
            return y
        idx = self.leaky_masks[name].reshape(-1).long()
        forced = x.reshape(-1)[idx].reshape(y.shape)
        self.pool_disagree[name] = float((y.detach() - forced.detach()).abs().max())
        self.pool_forced[name] = (int((y.detach() != forced.detach()).sum()), y.numel())
        return forced

    def cbl(self, x, name, stride=1, padding="same", act="leaky", bias=False):
        """conv -> BatchNormalization -> LeakyReLU(0.1) | Mish"""
        x = self.conv(x, f"{name}_conv", stride, padding, bias)
        x = self.bn(x, f"{name}_bn")
        if act != "leaky":
            out = L.mish(x)
        elif self.leaky_masks is not None and name in self.leaky_masks:
            m = self.leaky_masks[name]
            bad = (x.detach() > 0) != m
            self.mask_disagree[name] = float(x.detach().abs()[bad].max()) if bad.any() else 0.0
            self.mask_forced[name] = (int(bad.sum()), bad.numel())
            out = L.leaky_masked(x, m)
        else:
            out = L.leaky(x)
        if self.keep_acts:
            self.acts[name] = out.detach()
        return out


def _head_v234(c, x, i_out, anchors, version):
    """per anchor: xy sigmoid, wh exp*anchor, conf sigmoid, class sigmoid (v2: softmax); Concatenate."""
    outs = []
    for j, box in enumerate(anchors):
        p = f"out{i_out}_box{j + 1}"
        xy = torch.sigmoid(c.conv(x, f"{p}_xy_conv", bias=True))
        wh = torch.exp(c.conv(x, f"{p}_wh_conv", bias=True)) * torch.tensor(box, dtype=x.dtype)
        cf = torch.sigmoid(c.conv(x, f"{p}_conf_conv", bias=True))
        pr = c.conv(x, f"{p}_prob_conv", bias=True)
        pr = torch.softmax(pr, dim=-1) if version == 2 else torch.sigmoid(pr)
        outs += [xy, wh, cf, pr]
    return torch.cat(outs, dim=-1)


# ------------------------------------------- v3 -------------------------------------------------
def yolov3_forward(weights, x, anchors, training=False, unbiased_moving_var=False, leaky_masks=None):
    c = _Ctx(weights, training, x.dtype, unbiased_moving_var, leaky_masks)

    def resblock_body(t, blocks, name):
        t = c.cbl(t, f"{name}_dn", stride=2, padding="darknet_s2")
        for i in range(blocks):
            m = c.cbl(t, f"{name}_{i + 1}_1x1")
            m = c.cbl(m, f"{name}_{i + 1}_3x3")
            t = t + m
        return t

    def last_layers(t, name):
        for s in ("1_1x1", "1_3x3", "2_1x1", "2_3x3", "3_1x1"):
            t = c.cbl(t, f"{name}_{s}")
        return t, c.cbl(t, f"{name}_3_3x3")

    t = c.cbl(x, "conv1")
    t = resblock_body(t, 1, "block1")
    t = resblock_body(t, 2, "block2")
    t3 = resblock_body(t, 8, "block3")
    t4 = resblock_body(t3, 8, "block4")
    t5 = resblock_body(t4, 4, "block5")
    t, o1 = last_layers(t5, "last1")
    t = L.upsample2x(c.cbl(t, "up1"))
    t = torch.cat([t, t4], dim=-1)
    t, o2 = last_layers(t, "last2")
    t = L.upsample2x(c.cbl(t, "up2"))
    t = torch.cat([t, t3], dim=-1)
    t, o3 = last_layers(t, "last3")
    outs = []
    per = len(anchors) // 3
    for i, o in enumerate((o1, o2, o3)):
        outs.append(_head_v234(c, o, i + 1, anchors[i * per:(i + 1) * per], 3))
    return outs, c


def yolov3_tiny_forward(weights, x, anchors, training=False, unbiased_moving_var=False, leaky_masks=None):
    """tiny-YOLOv3 body, yolov3/models/darknet.py:107-135 (+ the heads of yolov3/models/__init__.py:13-70 on its two
    outputs, `create_model(backbone="tiny_darknet")`, yolov3/__init__.py:141-142). The reference's layers are
    unnamed; the names are this repo's (tf2_yolo_amd/graphs.py:_tiny_v3_body). MaxPooling2D(2, 'same'): stride 2
    down to 13x13, then ONE stride-1 pool that keeps 13x13 by padding bottom/right with -inf (:122)."""
    c = _Ctx(weights, training, x.dtype, unbiased_moving_var, leaky_masks)
    t = x
    for i, _f in enumerate((16, 32, 64, 128), start=1):
        t = c.cbl(t, f"tiny_c{i}")
        t = c.pool(t, f"tiny_p{i}", 2, 2, "same")
    t1 = c.cbl(t, "tiny_c5")
    t = c.pool(t1, "tiny_p5", 2, 2, "same")
    t = c.cbl(t, "tiny_c6")
    t = c.pool(t, "tiny_p6", 2, 1, "same")
    t = c.cbl(t, "tiny_c7")
    t2 = c.cbl(t, "tiny_c8")
    o1 = c.cbl(t2, "tiny_out1")
    t = L.upsample2x(c.cbl(t2, "tiny_up"))
    t = torch.cat([t, t1], dim=-1)
    o2 = c.cbl(t, "tiny_out2")
    per = len(anchors) // 2
    outs = [_head_v234(c, o, i + 1, anchors[i * per:(i + 1) * per], 3) for i, o in enumerate((o1, o2))]
    return outs, c


# ------------------------------------------- v4 -------------------------------------------------
def yolov4_forward(weights, x, anchors, training=False, unbiased_moving_var=False, leaky_masks=None):
    c = _Ctx(weights, training, x.dtype, unbiased_moving_var, leaky_masks)

    def resstage(t, blocks, name):
        t = c.cbl(t, f"{name}_dn", stride=2, padding="darknet_s2", act="mish")
        cross = c.cbl(t, f"{name}_cross", act="mish")
        t = c.cbl(t, f"{name}_pre", act="mish")
        for i in range(blocks):
            skip = t
            t = c.cbl(t, f"{name}_block{i + 1}_1x1", act="mish")
            t = c.cbl(t, f"{name}_block{i + 1}_3x3", act="mish")
            t = t + skip
        t = c.cbl(t, f"{name}_post", act="mish")
        t = torch.cat([t, cross], dim=-1)
        return c.cbl(t, f"{name}_out", act="mish")

    def last_layers(t, name):
        for s in ("1", "2", "3", "4", "5"):
            t = c.cbl(t, f"{name}_{s}")
        return t

    t = c.cbl(x, "conv1", act="mish")
    t = resstage(t, 1, "stage1")
    t = resstage(t, 2, "stage2")
    t3 = resstage(t, 8, "stage3")
    t4 = resstage(t3, 8, "stage4")
    t5 = resstage(t4, 4, "stage5")
    s = c.cbl(t5, "pan_td1_1")
    s = c.cbl(s, "pan_td1_2")
    s = c.cbl(s, "pan_td1_spp_pre")
    s = torch.cat([c.pool(s, "pan_td1_spp_pool1", 13, 1, "same"), c.pool(s, "pan_td1_spp_pool2", 9, 1, "same"),
                   c.pool(s, "pan_td1_spp_pool3", 5, 1, "same"), s], dim=-1)
    s = c.cbl(s, "pan_td1_3")
    s = c.cbl(s, "pan_td1_4")
    s = c.cbl(s, "pan_td1_5")
    s_up = L.upsample2x(c.cbl(s, "pan_td1_up"))
    m = c.cbl(t4, "pan_td2_pre")
    m = torch.cat([m, s_up], dim=-1)
    m = last_layers(m, "pan_td2")
    m_up = L.upsample2x(c.cbl(m, "pan_td2_up"))
    l = c.cbl(t3, "pan_td3_pre")
    l = torch.cat([l, m_up], dim=-1)
    l = last_layers(l, "pan_td3")
    out_l = c.cbl(l, "pan_out_l")
    l_dn = c.cbl(l, "pan_bu1_dn", stride=2, padding="darknet_s2")
    m = torch.cat([l_dn, m], dim=-1)
    m = last_layers(m, "pan_bu1")
    out_m = c.cbl(m, "pan_out_m")
    m_dn = c.cbl(m, "pan_bu2_dn", stride=2, padding="darknet_s2")
    s = torch.cat([m_dn, s], dim=-1)
    s = last_layers(s, "pan_bu2")
    out_s = c.cbl(s, "pan_out_s")
    outs = []
    per = len(anchors) // 3
    for i, o in enumerate((out_s, out_m, out_l)):
        outs.append(_head_v234(c, o, i + 1, anchors[i * per:(i + 1) * per], 4))
    return outs, c


# ------------------------------------------- v2 -------------------------------------------------
def yolov2_forward(weights, x, anchors, training=False, unbiased_moving_var=False, leaky_masks=None):
    c = _Ctx(weights, training, x.dtype, unbiased_moving_var, leaky_masks)

    def cbl(t, name):
        return c.cbl(t, name, bias=True)

    t = cbl(x, "conv1")
    t = c.pool(t, "pool1", 2, 2)
    t = cbl(t, "conv2")
    t = c.pool(t, "pool2", 2, 2)
    for n in ("conv3_1", "conv3_2", "conv3_3"):
        t = cbl(t, n)
    t = c.pool(t, "pool3", 2, 2)
    for n in ("conv4_1", "conv4_2", "conv4_3"):
        t = cbl(t, n)
    t = c.pool(t, "pool4", 2, 2)
    for n in ("conv5_1", "conv5_2", "conv5_3", "conv5_4", "conv5_5"):
        t = cbl(t, n)
    passthrough = t
    t = c.pool(t, "pool5", 2, 2)
    for n in ("conv6_1", "conv6_2", "conv6_3", "conv6_4", "conv6_5", "conv7_1", "conv7_2"):
        t = cbl(t, n)
    p = L.space_to_depth2(cbl(passthrough, "passthrough_conv"))
    t = cbl(torch.cat([p, t], dim=-1), "conv8")
    return [_head_v234(c, t, 1, anchors, 2)], c


# ------------------------------------------- v1.5 -----------------------------------------------
def yolov1_5_forward(weights, x, training=False, unbiased_moving_var=False, leaky_masks=None):
    c = _Ctx(weights, training, x.dtype, unbiased_moving_var, leaky_masks)

    def cbl(t, name, stride=1):
        return c.cbl(t, name, stride=stride, bias=True)

    t = cbl(x, "conv1", 2)
    t = c.pool(t, "pool1", 2, 2)
    t = cbl(t, "conv2")
    t = c.pool(t, "pool2", 2, 2)
    for n in ("conv3_1", "conv3_2", "conv3_3", "conv3_4"):
        t = cbl(t, n)
    t = c.pool(t, "pool3", 2, 2)
    for i in range(1, 10):
        t = cbl(t, f"conv4_{i}")
    t = c.pool(t, "pool4", 2, 2)
    for n in ("conv5_1", "conv5_2", "conv5_3", "conv5_4", "conv5_5"):
        t = cbl(t, n)
    t = cbl(t, "conv5_6", 2)
    t = cbl(t, "conv6_1")
    t = cbl(t, "conv6_2")
    xywhc = torch.sigmoid(c.conv(t, "out1_xywhc_conv", bias=True))
    prob = torch.softmax(c.conv(t, "out1_prob_conv", bias=True), dim=-1)
    return [torch.cat([xywhc, prob], dim=-1)], c
