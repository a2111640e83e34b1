"""`yolo_body` / `tiny_yolo_body` / `yolo_head` with the reference's signatures, as graph-builder entry points.

  v3   yolov3/models/darknet.py:71-135 (yolo_body, tiny_yolo_body), yolov3/models/__init__.py:13-70 (yolo_head)
  v4   yolov4/models/darknet.py:72-146, yolov4/models/__init__.py:14-71
  v2   yolov2/models/darknet.py:32-106
  v1.5 yolov1_5/models/darknet.py:26-55

The reference returns tf.keras Models built from Keras tensors. Here `yolo_body` returns a `BodyModel`: the SYMBOLIC
body graph (a GraphBuilder without head units; `.input`, `.output`, `.input_shape`, `.output_shape` describe it),
and `yolo_head(model_body, ...)` appends the fused head units and hands back the executable `Model` that
`Yolo.create_model` stores in `yolo.model` -- the same two-call sequence as yolov3/__init__.py:122-175. Weights given
to a body (`set_weights`, `pretrained_darknet`, a `.npz` path as `pretrained_weights`) are applied when the head
creates the device buffers. Keras-tensor surgery (layers[i].output, Model(inputs, outputs)) is not available: the
graphs are fixed by graphs.py, which mirrors the reference's definitions layer for layer.
"""
from . import graphs
from .model import Model


def _offline(what, value):
    raise ValueError(f"{what}={value!r} needs a network download, which is unavailable; "
                     "pass None (random init) or the path of a .npz weight file")


class BodyModel:
    """Symbolic YOLO body: what `yolo_body` returns and `yolo_head` consumes."""

    def __init__(self, builder, outs, version):
        self.builder, self._outs, self.version = builder, list(outs), version
        self._pending = []     # ("list", weights) | ("model", Model) | ("file", path), applied by yolo_head in order

    @property
    def input(self):
        return self.builder.input

    @property
    def input_shape(self):
        return self.builder.input.shape

    @property
    def output(self):
        return self._outs[0] if self.version in (1, 2) else list(self._outs)

    @property
    def output_shape(self):
        shapes = [t.shape for t in self._outs]
        return shapes[0] if self.version in (1, 2) else shapes

    def layer_names(self):
        names = []
        for u in self.builder.units:
            if u.kind == "conv":
                names.append(f"{u.name}_conv")
                if u.bn:
                    names.append(f"{u.name}_bn")
            elif u.kind != "head":
                names.append(u.name)
        return names

    def count_params(self):
        from .engine import count_params
        return sum(count_params(self.builder))

    def set_weights(self, weights):
        """Keras order (layer by layer: kernel[, bias], gamma, beta, moving mean, moving variance)."""
        self._pending.append(("list", list(weights)))

    def load_weights(self, path, by_name=False, skip_mismatch=False):
        self._pending.append(("file", (path, by_name, skip_mismatch)))

    def get_weights(self):
        raise ValueError("a body graph is symbolic until yolo_head() creates the model: read the weights from the "
                         "Model it returns (model.get_weights() / model.get_layer(name).get_weights())")

    def _apply(self, model):
        body_names = [n for n in model.layer_names() if n in set(self.layer_names())]
        for kind, payload in self._pending:
            if kind == "model":
                model.set_body_weights(payload)
            elif kind == "file":
                path, by_name, skip_mismatch = payload
                model.load_weights(path, by_name=True, skip_mismatch=skip_mismatch)
            else:
                i = 0
                for n in body_names:
                    layer = model.get_layer(n)
                    k = len(layer.get_weights())
                    if k:
                        if i + k > len(payload):
                            raise ValueError(f"set_weights: the list has {len(payload)} arrays, the body needs more")
                        layer.set_weights(payload[i:i + k])
                        i += k
                if i != len(payload):
                    raise ValueError(f"set_weights: the body has {i} weight arrays, the list {len(payload)}")


def _body(builder_outs, version, pretrained, pretrained_weights, what="pretrained_darknet"):
    body = BodyModel(*builder_outs, version)
    if pretrained is not None:
        if not hasattr(pretrained, "get_layer"):
            raise ValueError(f"{what} must be a model created by this package (layers are matched by name); the "
                             "ImageNet classifier variants are outside the HIP path (SURVEY.md section 2 row 16)")
        body._pending.append(("model", pretrained))
    if pretrained_weights is not None:
        if pretrained_weights in ("pascal_voc", "ms_coco", "imagenet"):
            _offline("pretrained_weights", pretrained_weights)
        body.load_weights(pretrained_weights)
    return body


def _model(model_body, version, seed, bn_unbiased_moving_var):
    m = Model(model_body.builder, version=version, seed=seed, unbiased_moving_var=bn_unbiased_moving_var)
    model_body._apply(m)
    return m


# ---- YOLOv3 ---------------------------------------------------------------------------------------------
def yolo_body_v3(input_shape=(416, 416, 3), pretrained_darknet=None, pretrained_weights=None):
    """Create YOLO_V3 model CNN body (yolov3/models/darknet.py:71-104)."""
    return _body(graphs.yolov3_body(input_shape, "full_darknet"), 3, pretrained_darknet, pretrained_weights)


def tiny_yolo_body(input_shape=(416, 416, 3)):
    """Create Tiny YOLO_v3 model CNN body (yolov3/models/darknet.py:107-135)."""
    return BodyModel(*graphs.yolov3_body(input_shape, "tiny_darknet"), 3)


def yolo_head_v3(model_body, class_num=10, anchors=graphs.V3_DEFAULT_ANCHORS, seed=1234, bn_unbiased_moving_var=True):
    """YOLOv3 head (yolov3/models/__init__.py:13-70)."""
    graphs.fpn_head(model_body.builder, model_body._outs, class_num, [list(a) for a in anchors], 3)
    return _model(model_body, 3, seed, bn_unbiased_moving_var)


# ---- YOLOv4 ---------------------------------------------------------------------------------------------
def yolo_body_v4(input_shape=(608, 608, 3), pretrained_darknet=None, pretrained_weights=None):
    """Create YOLOv4 body (yolov4/models/darknet.py:72-146)."""
    return _body(graphs.yolov4_body(input_shape), 4, pretrained_darknet, pretrained_weights)


def yolo_head_v4(model_body, class_num=80, anchors=graphs.V4_DEFAULT_ANCHORS, seed=1234, bn_unbiased_moving_var=True):
    """YOLOv4 head (yolov4/models/__init__.py:14-71)."""
    graphs.fpn_head(model_body.builder, model_body._outs, class_num, [list(a) for a in anchors], 4)
    return _model(model_body, 4, seed, bn_unbiased_moving_var)


# ---- YOLOv2 ---------------------------------------------------------------------------------------------
V2_HEAD_DEFAULT_ANCHORS = [(0.04405615, 0.05210654), (0.14418923, 0.15865615), (0.25680231, 0.42110308),
                           (0.60637077, 0.27136769), (0.75157846, 0.70525231)]   # yolov2/models/darknet.py:69-73


def yolo_body_v2(input_shape=(416, 416, 3), backbone="darknet", pretrained_backbone=None):
    """Body of YOLOv2 (yolov2/models/darknet.py:32-65)."""
    if backbone != "darknet":
        if backbone in ("unet", "mobilenet"):
            raise ValueError(f"backbone {backbone!r} is outside the HIP path (SURVEY.md section 2 row 15)")
        raise ValueError(f"Invalid backbone: {backbone}")
    return _body(graphs.yolov2_body(input_shape), 2, pretrained_backbone, None, "pretrained_backbone")


def yolo_head_v2(model_body, class_num=10, anchors=V2_HEAD_DEFAULT_ANCHORS, seed=1234, bn_unbiased_moving_var=True):
    """Head of YOLOv2 (yolov2/models/darknet.py:68-106)."""
    anchors = [list(a) for a in anchors]
    model_body.builder.head(model_body._outs[0], len(anchors), class_num, 2, anchors, "out1", level=0)
    return _model(model_body, 2, seed, bn_unbiased_moving_var)


# ---- YOLOv1.5 -------------------------------------------------------------------------------------------
def yolo_body_v1(input_shape=(448, 448, 3), pretrained_darknet=None):
    """Body of YOLOv1 (yolov1_5/models/darknet.py:26-34)."""
    return _body(graphs.yolov1_5_body(input_shape), 1, pretrained_darknet, None)


def yolo_head_v1(model_body, bbox_num=2, class_num=10, seed=1234, bn_unbiased_moving_var=True):
    """Head of YOLOv1 (yolov1_5/models/darknet.py:37-55)."""
    model_body.builder.head(model_body._outs[0], bbox_num, class_num, 1, None, "out1", level=0)
    return _model(model_body, 1, seed, bn_unbiased_moving_var)


def _keras_tensor_api(name):
    def f(*args, **kwargs):
        raise NotImplementedError(
            f"{name} builds tf.keras layers / keras.applications backbones; the HIP executor runs the fixed graphs of "
            "tf2_yolo_amd/graphs.py (SURVEY.md section 2 rows 15-16: out of scope). Use yolo_body(...) + yolo_head(...)")
    f.__name__ = name
    return f


yolo_keras_app_body = _keras_tensor_api("yolo_keras_app_body")
darknet53 = _keras_tensor_api("darknet53")
csp_darknet53 = _keras_tensor_api("csp_darknet53")
darknet19 = _keras_tensor_api("darknet19")
darknet = _keras_tensor_api("darknet")
