"""Model graphs of the four YOLO versions, built for the HIP executor (engine.py).

Each function mirrors one reference graph definition layer for layer (names included, so
weights can be exchanged by layer name) but emits fused units:
  conv + BatchNormalization + LeakyReLU/Mish (+ residual Add)  -> one ConvUnit
  the 4*A per-anchor head convs + activations + Concatenate    -> one HeadUnit
"""
from ._lib import ACT_LEAKY, ACT_LINEAR, ACT_MISH
from .engine import GraphBuilder, he_normal_krsc, random_normal_002

V3_DEFAULT_ANCHORS = [[0.89663461, 0.78365384], [0.37500000, 0.47596153], [0.27884615, 0.21634615],
                      [0.14182692, 0.28605769], [0.14903846, 0.10817307], [0.07211538, 0.14663461],
                      [0.07932692, 0.05528846], [0.03846153, 0.07211538], [0.02403846, 0.03125000]]
V4_DEFAULT_ANCHORS = [[0.75493421, 0.65953947], [0.31578947, 0.39967105], [0.23355263, 0.18092105],
                      [0.11842105, 0.24013158], [0.12500000, 0.09046053], [0.05921053, 0.12335526],
                      [0.06578947, 0.04605263], [0.03125000, 0.05921053], [0.01973684, 0.02631579]]


# ---------------------------------------------------------------------------------------------
# YOLOv3 (yolov3/models/backbone.py:58-95, yolov3/models/darknet.py:71-104,
#         yolov3/models/__init__.py:13-70)
# ---------------------------------------------------------------------------------------------
def _v3_resblock_body(b, x, filters, blocks, name):
    # ZeroPadding2D(((1,0),(1,0))) + 3x3 stride-2 'valid' conv  (backbone.py:61-63)
    x = b.conv(x, filters, 3, f"{name}_dn", stride=2, padding="darknet_s2")
    for i in range(blocks):
        y = b.conv(x, filters // 2, 1, f"{name}_{i + 1}_1x1")
        x = b.conv(y, filters, 3, f"{name}_{i + 1}_3x3", residual=x)  # Add fused (backbone.py:71)
    return x


def darknet53_body(b, x):
    """backbone.py:74-82. Returns (block3 out, block4 out, block5 out)."""
    x = b.conv(x, 32, 3, "conv1")
    x = _v3_resblock_body(b, x, 64, 1, "block1")
    x = _v3_resblock_body(b, x, 128, 2, "block2")
    x3 = _v3_resblock_body(b, x, 256, 8, "block3")     # Keras layer index 92
    x4 = _v3_resblock_body(b, x3, 512, 8, "block4")    # Keras layer index 152
    x5 = _v3_resblock_body(b, x4, 1024, 4, "block5")
    return x3, x4, x5


def _v3_last_layers(b, x, filters, name):
    """backbone.py:85-95"""
    x = b.conv(x, filters, 1, f"{name}_1_1x1")
    x = b.conv(x, filters * 2, 3, f"{name}_1_3x3")
    x = b.conv(x, filters, 1, f"{name}_2_1x1")
    x = b.conv(x, filters * 2, 3, f"{name}_2_3x3")
    x = b.conv(x, filters, 1, f"{name}_3_1x1")
    out = b.conv(x, filters * 2, 3, f"{name}_3_3x3")
    return x, out


def yolov3_body(input_shape, backbone="full_darknet"):
    """yolo_body / tiny_yolo_body (yolov3/models/darknet.py:71-135): (builder, [out1, out2(, out3)]) coarse -> fine."""
    b = GraphBuilder(input_shape, kernel_init=he_normal_krsc)
    if backbone == "full_darknet":
        x3, x4, x5 = darknet53_body(b, b.input)
        x, out1 = _v3_last_layers(b, x5, 512, "last1")
        x = b.conv(x, 256, 1, "up1")
        x = b.upsample(x, "up1_up")
        x = b.concat([x, x4], "concat1")
        x, out2 = _v3_last_layers(b, x, 256, "last2")
        x = b.conv(x, 128, 1, "up2")
        x = b.upsample(x, "up2_up")
        x = b.concat([x, x3], "concat2")
        x, out3 = _v3_last_layers(b, x, 128, "last3")
        body_outs = [out1, out2, out3]
    elif backbone == "tiny_darknet":
        body_outs = _tiny_v3_body(b)
    else:
        raise ValueError(f"Invalid backbone: {backbone}")
    return b, body_outs


def fpn_head(b, body_outs, class_num, anchors, version):
    """yolo_head of v3 / v4 (yolov3/models/__init__.py:13-70, yolov4/models/__init__.py:14-71)."""
    tensor_num = len(body_outs)
    if len(anchors) % tensor_num > 0:
        raise ValueError("The total number of anchor boxs should be a multiple of the number "
                         f"{tensor_num} of output tensors")
    abox = len(anchors) // tensor_num
    for i, t in enumerate(body_outs):
        b.head(t, abox, class_num, version, anchors[i * abox:(i + 1) * abox], f"out{i + 1}", level=i)
    return b


def build_yolov3(input_shape, class_num, anchors=None, backbone="full_darknet"):
    anchors = V3_DEFAULT_ANCHORS if anchors is None else anchors
    b, body_outs = yolov3_body(input_shape, backbone)
    return fpn_head(b, body_outs, class_num, anchors, 3)


def _tiny_v3_body(b):
    """yolov3/models/darknet.py:107-135 (unnamed layers in the reference; named tiny_* here)."""
    x = b.conv(b.input, 16, 3, "tiny_c1")
    x = b.maxpool(x, 2, "tiny_p1", stride=2, padding="same")
    x = b.conv(x, 32, 3, "tiny_c2")
    x = b.maxpool(x, 2, "tiny_p2", stride=2, padding="same")
    x = b.conv(x, 64, 3, "tiny_c3")
    x = b.maxpool(x, 2, "tiny_p3", stride=2, padding="same")
    x = b.conv(x, 128, 3, "tiny_c4")
    x = b.maxpool(x, 2, "tiny_p4", stride=2, padding="same")
    t1 = b.conv(x, 256, 3, "tiny_c5")
    x = b.maxpool(t1, 2, "tiny_p5", stride=2, padding="same")
    x = b.conv(x, 512, 3, "tiny_c6")
    x = b.maxpool(x, 2, "tiny_p6", stride=1, padding="same")
    x = b.conv(x, 1024, 3, "tiny_c7")
    t2 = b.conv(x, 256, 1, "tiny_c8")
    out1 = b.conv(t2, 512, 3, "tiny_out1")
    x = b.conv(t2, 128, 1, "tiny_up")
    x = b.upsample(x, "tiny_up_up")
    x = b.concat([x, t1], "tiny_concat")
    out2 = b.conv(x, 256, 3, "tiny_out2")
    return [out1, out2]


# ---------------------------------------------------------------------------------------------
# YOLOv4 (yolov4/models/backbone.py:113-185, yolov4/models/darknet.py:72-146,
#         yolov4/models/__init__.py:14-71)
# ---------------------------------------------------------------------------------------------
def _v4_resstage(b, x, filters, blocks, narrow, name):
    mid = filters // 2 if narrow else filters
    x = b.conv(x, filters, 3, f"{name}_dn", stride=2, padding="darknet_s2", act=ACT_MISH)
    cross = b.conv(x, mid, 1, f"{name}_cross", act=ACT_MISH)
    x = b.conv(x, mid, 1, f"{name}_pre", act=ACT_MISH)
    for i in range(blocks):
        y = b.conv(x, filters // 2, 1, f"{name}_block{i + 1}_1x1", act=ACT_MISH)
        x = b.conv(y, mid, 3, f"{name}_block{i + 1}_3x3", act=ACT_MISH, residual=x)
    x = b.conv(x, mid, 1, f"{name}_post", act=ACT_MISH)
    x = b.concat([x, cross], f"{name}_concat")
    return b.conv(x, filters, 1, f"{name}_out", act=ACT_MISH)


def csp_darknet53_body(b, x):
    x = b.conv(x, 32, 3, "conv1", act=ACT_MISH)
    x = _v4_resstage(b, x, 64, 1, False, "stage1")
    x = _v4_resstage(b, x, 128, 2, True, "stage2")
    x3 = _v4_resstage(b, x, 256, 8, True, "stage3")    # Keras layer index 131
    x4 = _v4_resstage(b, x3, 512, 8, True, "stage4")   # Keras layer index 204
    x5 = _v4_resstage(b, x4, 1024, 4, True, "stage5")
    return x3, x4, x5


def _v4_last_layers(b, x, filters, name):
    x = b.conv(x, filters, 1, f"{name}_1")
    x = b.conv(x, filters * 2, 3, f"{name}_2")
    x = b.conv(x, filters, 1, f"{name}_3")
    x = b.conv(x, filters * 2, 3, f"{name}_4")
    return b.conv(x, filters, 1, f"{name}_5")


def yolov4_body(input_shape):
    """yolo_body (yolov4/models/darknet.py:72-146): (builder, [out_s, out_m, out_l])."""
    b = GraphBuilder(input_shape, kernel_init=random_normal_002)
    x3, x4, x5 = csp_darknet53_body(b, b.input)
    s = b.conv(x5, 512, 1, "pan_td1_1")
    s = b.conv(s, 1024, 3, "pan_td1_2")
    s = b.conv(s, 512, 1, "pan_td1_spp_pre")
    # spp_module (backbone.py:176-185): concat(pool13, pool9, pool5, x), stride 1 'same'
    p1 = b.maxpool(s, 13, "pan_td1_spp_pool1", stride=1, padding="same")
    p2 = b.maxpool(s, 9, "pan_td1_spp_pool2", stride=1, padding="same")
    p3 = b.maxpool(s, 5, "pan_td1_spp_pool3", stride=1, padding="same")
    s = b.concat([p1, p2, p3, s], "pan_td1_spp_concat")
    s = b.conv(s, 512, 1, "pan_td1_3")
    s = b.conv(s, 1024, 3, "pan_td1_4")
    s = b.conv(s, 512, 1, "pan_td1_5")
    s_up = b.conv(s, 256, 1, "pan_td1_up")
    s_up = b.upsample(s_up, "pan_td1_up_up")
    m = b.conv(x4, 256, 1, "pan_td2_pre")
    m = b.concat([m, s_up], "pan_td1_concat")
    m = _v4_last_layers(b, m, 256, "pan_td2")
    m_up = b.conv(m, 128, 1, "pan_td2_up")
    m_up = b.upsample(m_up, "pan_td2_up_up")
    l = b.conv(x3, 128, 1, "pan_td3_pre")
    l = b.concat([l, m_up], "pan_td2_concat")
    l = _v4_last_layers(b, l, 128, "pan_td3")
    out_l = b.conv(l, 256, 3, "pan_out_l")
    l_dn = b.conv(l, 256, 3, "pan_bu1_dn", stride=2, padding="darknet_s2")
    m = b.concat([l_dn, m], "pan_bu1_concat")
    m = _v4_last_layers(b, m, 256, "pan_bu1")
    out_m = b.conv(m, 512, 3, "pan_out_m")
    m_dn = b.conv(m, 512, 3, "pan_bu2_dn", stride=2, padding="darknet_s2")
    s = b.concat([m_dn, s], "pan_bu2_concat")
    s = _v4_last_layers(b, s, 512, "pan_bu2")
    out_s = b.conv(s, 1024, 3, "pan_out_s")
    return b, [out_s, out_m, out_l]


def build_yolov4(input_shape, class_num, anchors=None):
    anchors = V4_DEFAULT_ANCHORS if anchors is None else anchors
    b, body_outs = yolov4_body(input_shape)
    return fpn_head(b, body_outs, class_num, anchors, 4)


# ---------------------------------------------------------------------------------------------
# YOLOv2 (yolov2/models/backbone.py:42-73, yolov2/models/darknet.py:32-106)
# ---------------------------------------------------------------------------------------------
def yolov2_body(input_shape):
    """yolo_body, backbone "darknet" (yolov2/models/darknet.py:32-65): (builder, [out])."""
    b = GraphBuilder(input_shape, kernel_init=he_normal_krsc)

    def cbl(x, f, k, name):
        return b.conv(x, f, k, name, bias=True)   # Conv2D(use_bias default True)+BN+Leaky (backbone.py:11-18)

    x = cbl(b.input, 32, 3, "conv1")
    x = b.maxpool(x, 2, "pool1")
    x = cbl(x, 64, 3, "conv2")
    x = b.maxpool(x, 2, "pool2")
    x = cbl(x, 128, 3, "conv3_1")
    x = cbl(x, 64, 1, "conv3_2")
    x = cbl(x, 128, 3, "conv3_3")
    x = b.maxpool(x, 2, "pool3")
    x = cbl(x, 256, 3, "conv4_1")
    x = cbl(x, 128, 1, "conv4_2")
    x = cbl(x, 256, 3, "conv4_3")
    x = b.maxpool(x, 2, "pool4")
    x = cbl(x, 512, 3, "conv5_1")
    x = cbl(x, 256, 1, "conv5_2")
    x = cbl(x, 512, 3, "conv5_3")
    x = cbl(x, 256, 1, "conv5_4")
    passthrough = cbl(x, 512, 3, "conv5_5")          # Keras layer index 43 (26x26x512)
    x = b.maxpool(passthrough, 2, "pool5")
    x = cbl(x, 1024, 3, "conv6_1")
    x = cbl(x, 512, 1, "conv6_2")
    x = cbl(x, 1024, 3, "conv6_3")
    x = cbl(x, 512, 1, "conv6_4")
    x = cbl(x, 1024, 3, "conv6_5")
    # yolo_body (darknet.py:32-65)
    x = cbl(x, 1024, 3, "conv7_1")
    x = cbl(x, 1024, 3, "conv7_2")
    p = cbl(passthrough, 64, 3, "passthrough_conv")
    p = b.space_to_depth(p, "passthrough_s2d")
    x = b.concat([p, x], "passthrough_concat")
    x = cbl(x, 1024, 3, "conv8")
    return b, [x]


def build_yolov2(input_shape, class_num, anchors):
    b, (x,) = yolov2_body(input_shape)
    b.head(x, len(anchors), class_num, 2, anchors, "out1", level=0)   # yolo_head, yolov2/models/darknet.py:68-106
    return b


# ---------------------------------------------------------------------------------------------
# YOLOv1.5 (yolov1_5/models/backbone.py:18-48, yolov1_5/models/darknet.py:26-55)
# ---------------------------------------------------------------------------------------------
def yolov1_5_body(input_shape):
    """yolo_body (yolov1_5/models/darknet.py:26-34): (builder, [out])."""
    b = GraphBuilder(input_shape, kernel_init=he_normal_krsc)

    def cbl(x, f, k, name, stride=1):
        return b.conv(x, f, k, name, stride=stride, padding="same", bias=True)

    x = cbl(b.input, 64, 7, "conv1", stride=2)
    x = b.maxpool(x, 2, "pool1")
    x = cbl(x, 192, 3, "conv2")
    x = b.maxpool(x, 2, "pool2")
    x = cbl(x, 128, 1, "conv3_1")
    x = cbl(x, 256, 3, "conv3_2")
    x = cbl(x, 256, 1, "conv3_3")
    x = cbl(x, 512, 3, "conv3_4")
    x = b.maxpool(x, 2, "pool3")
    for i in range(4):
        x = cbl(x, 256, 1, f"conv4_{2 * i + 1}")
        x = cbl(x, 512, 3, f"conv4_{2 * i + 2}")
    x = cbl(x, 1024, 3, "conv4_9")
    x = b.maxpool(x, 2, "pool4")
    x = cbl(x, 512, 1, "conv5_1")
    x = cbl(x, 1024, 3, "conv5_2")
    x = cbl(x, 512, 1, "conv5_3")
    x = cbl(x, 1024, 3, "conv5_4")
    x = cbl(x, 1024, 3, "conv5_5")
    x = cbl(x, 1024, 3, "conv5_6", stride=2)
    x = cbl(x, 1024, 3, "conv6_1")
    x = cbl(x, 1024, 3, "conv6_2")
    return b, [x]


def build_yolov1_5(input_shape, class_num, bbox_num=2):
    b, (x,) = yolov1_5_body(input_shape)
    b.head(x, bbox_num, class_num, 1, None, "out1", level=0)          # yolo_head, yolov1_5/models/darknet.py:37-55
    return b
