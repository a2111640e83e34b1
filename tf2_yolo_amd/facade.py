"""`Yolo` helper classes with the reference's constructor / create_model / loss / metrics surface
(argument names, defaults, error behaviour), backed by the HIP executor.

  v3   yolov3/__init__.py:56-493        v4   yolov4/__init__.py:56-592
  v2   yolov2/__init__.py:29-369        v1.5 yolov1_5/__init__.py:29-347

File readers and matplotlib visualisation (read_file_to_dataset / read_file_to_sequence /
vis_img) are outside the accelerated path (host file I/O that needs
cv2 / imgaug / bs4, absent here) and raise NotImplementedError with that explanation.
Pretrained downloads are impossible offline: pretrained_body / pretrained_weights strings that
name a download ("pascal_voc", "ms_coco", "imagenet") raise; pass None or a .npz path.
"""
from collections.abc import Iterable

import numpy as np

from . import bodies, graphs, losses


class MetricKind(object):
    """names of metric kind (yolov3/__init__.py:33-38)"""
    obj_acc = "obj_acc"
    mean_iou = "mean_iou"
    class_acc = "class_acc"
    recall = "recall"


def _parse_recall_threshold(kind):
    """The reference's mini grammar: text after 'recall' up to the last '+' (yolov3/__init__.py:474-483)."""
    iou_threshold = kind[kind.find("recall") + 6:]
    end = iou_threshold.rfind("+")
    if end < 0:
        end = None
    iou_threshold = iou_threshold[:end]
    return 0.5 if iou_threshold == "" else float(iou_threshold)


def _metric_list(kind, version, grid_shape, bbox_num, class_num):
    out = []
    if "obj" in kind:
        out.append(losses.wrap_obj_acc(grid_shape, bbox_num, class_num, version=version))
    if "iou" in kind:
        out.append(losses.wrap_mean_iou(grid_shape, bbox_num, class_num, version=version))
    if "class" in kind:
        out.append(losses.wrap_class_acc(grid_shape, bbox_num, class_num, version=version))
    if "recall" in kind:
        out.append(losses.wrap_recall(grid_shape, bbox_num, class_num, iou_threshold=_parse_recall_threshold(kind),
                                      version=version))
    return out


def _loss_weight_list(loss_weight, keys):
    if isinstance(loss_weight, dict):
        return [loss_weight[k] for k in keys]
    return loss_weight


def _offline(what, value):
    raise ValueError(f"{what}={value!r} needs a network download, which is unavailable; "
                     "pass None (random init) or the path of a .npz weight file")


class _IOUnavailable:
    def _io(self, name):
        raise NotImplementedError(
            f"{name}: dataset reading / plotting is host file I/O outside the accelerated hot path "
            "(needs cv2, imgaug, bs4 which are not installed); feed ndarrays to model.fit / model.predict "
            "and use tf2_yolo_amd.tools.decode / nms for post-processing")

    def read_file_to_dataset(self, *a, **k):
        self._io("read_file_to_dataset")

    def read_file_to_sequence(self, *a, **k):
        self._io("read_file_to_sequence")

    def vis_img(self, *a, **k):
        self._io("vis_img")


# ---------------------------------------------------------------------------------------------
class YoloV3(_IOUnavailable):
    def __init__(self, input_shape=(416, 416, 3), class_names: list = []):
        self.input_shape = input_shape
        self.grid_shape = input_shape[0] // 32, input_shape[1] // 32
        self.abox_num = 3
        self.class_names = class_names
        self.class_num = len(class_names)
        self.fpn_layers = 3
        self.anchors = None
        self.model = None
        self.file_names = None

    def create_model(self, anchors=graphs.V3_DEFAULT_ANCHORS, backbone="full_darknet", pretrained_weights=None,
                     pretrained_body="pascal_voc", seed=1234, bn_unbiased_moving_var=True):
        """yolov3/__init__.py:100-181. Extra keyword (no reference counterpart): bn_unbiased_moving_var -- which batch
        variance BatchNormalization feeds into its moving average. True (default) = the Bessel-corrected variance that
        tf.keras' fused BatchNormalization of TF 2.0 - 2.15 (the reference's `tensorflow>=2.0.0` era) feeds; False =
        the biased one (non-fused / Keras 3 behaviour). Normalisation itself always uses the biased batch variance."""
        if isinstance(pretrained_body, str):
            _offline("pretrained_body", pretrained_body)
        if backbone not in ("full_darknet", "tiny_darknet"):
            if backbone in ("resnet50", "resnet101", "resnet152", "resnet50v2", "resnet101v2", "resnet152v2"):
                raise ValueError(f"backbone {backbone!r} lives in keras.applications, outside the HIP path "
                                 "(SURVEY.md section 2 row 15)")
            raise ValueError(f"Invalid backbone: {backbone}")
        # the reference's two calls (yolov3/__init__.py:122-175): yolo_body / tiny_yolo_body, then yolo_head
        if backbone == "full_darknet":
            model_body = bodies.yolo_body_v3(self.input_shape, pretrained_darknet=pretrained_body)
        else:
            model_body = bodies.tiny_yolo_body(self.input_shape)
            if pretrained_body is not None:
                model_body._pending.append(("model", pretrained_body))
        self.model = bodies.yolo_head_v3(model_body, self.class_num, anchors, seed=seed,
                                         bn_unbiased_moving_var=bn_unbiased_moving_var)
        if pretrained_weights is not None:
            self.model.load_weights(pretrained_weights)
        self.anchors = anchors
        self.grid_shape = tuple(self.model.output[0].shape[1:3])
        self.fpn_layers = len(self.model.output)
        self.abox_num = len(self.anchors) // self.fpn_layers

    def loss(self, binary_weight=1, loss_weight=[1, 1, 5, 1], ignore_thresh=.6, use_focal_loss=False,
             focal_loss_gamma=2, use_scale=True):
        if not isinstance(binary_weight, Iterable) or len(binary_weight) != self.fpn_layers:
            binary_weight = [binary_weight] * self.fpn_layers
        loss_weight = _loss_weight_list(loss_weight, ("xy", "wh", "conf", "prob"))
        loss_list = []
        for fpn_id in range(self.fpn_layers):
            amp = 2 ** fpn_id
            grid_shape = (self.grid_shape[0] * amp, self.grid_shape[1] * amp)
            aid = self.abox_num * fpn_id
            loss_list.append(losses.wrap_yolo_loss_v3(
                grid_shape=grid_shape, bbox_num=self.abox_num, class_num=self.class_num,
                anchors=self.anchors[aid:aid + self.abox_num], binary_weight=binary_weight[fpn_id],
                loss_weight=loss_weight, ignore_thresh=ignore_thresh, use_focal_loss=use_focal_loss,
                focal_loss_gamma=focal_loss_gamma, use_scale=use_scale))
        return loss_list

    def metrics(self, kind="obj_acc"):
        out = []
        for fpn_id in range(self.fpn_layers):
            amp = 2 ** fpn_id
            out.append(_metric_list(kind, 3, (self.grid_shape[0] * amp, self.grid_shape[1] * amp), self.abox_num,
                                    self.class_num))
        return out


# ---------------------------------------------------------------------------------------------
class YoloV4(_IOUnavailable):
    def __init__(self, input_shape=(608, 608, 3), class_names: list = []):
        self.input_shape = input_shape
        self.grid_shape = input_shape[0] // 32, input_shape[1] // 32
        self.abox_num = 3
        self.class_names = class_names
        self.class_num = len(class_names)
        self.pan_layers = 3
        self._model = None
        self._file_names = None
        self._anchors_trainable = False

    @property
    def model(self):
        if self._model is None:
            raise ValueError("You haven't created a model by using create_model().")
        return self._model

    @model.setter
    def model(self, _):
        raise ValueError("Can't set attribute directly, please create a model by using create_model().")

    @model.deleter
    def model(self):
        self._model = None

    def _anchor_layer(self, i_out, i_box):
        return self.model.get_layer(name=f"out{i_out + 1}_box{i_box + 1}_anchor")

    @property
    def anchors(self):
        if self._model is None:
            raise ValueError("To get anchors, you have to create a model first.")
        rows = [self._anchor_layer(i, j).get_weights()[0] for i in range(self.pan_layers) for j in range(self.abox_num)]
        return np.squeeze(np.vstack(rows)).tolist()

    @anchors.setter
    def anchors(self, anchor_boxes):
        for i_out in range(self.pan_layers):
            start = i_out * self.abox_num
            for i_box, box in enumerate(anchor_boxes[start:start + self.abox_num]):
                self._anchor_layer(i_out, i_box).set_weights([np.expand_dims(box, axis=(0, 1, 2))])

    @property
    def anchors_trainable(self):
        return self._anchors_trainable

    @anchors_trainable.setter
    def anchors_trainable(self, trainable):
        for i_out in range(self.pan_layers):
            for i_box in range(self.abox_num):
                self._anchor_layer(i_out, i_box).trainable = trainable
        self.model.set_anchors_trainable(bool(trainable))
        self._anchors_trainable = trainable

    @property
    def file_names(self):
        if self._file_names is None:
            raise ValueError("You haven't read files.")
        return self._file_names

    def reshape_anchors(self, ori_shape, shape=None):
        if shape is None:
            shape = self.input_shape[1::-1]
        amp = ori_shape[0] / shape[0], ori_shape[1] / shape[1]
        for i_out in range(self.pan_layers):
            for i_box in range(self.abox_num):
                layer = self._anchor_layer(i_out, i_box)
                layer.set_weights([layer.get_weights()[0] * amp])

    def create_model(self, anchors=None, backbone="csp_darknet", pretrained_weights=None, pretrained_body="ms_coco",
                     seed=1234, bn_unbiased_moving_var=True):
        use_arg_anchors = True
        if pretrained_weights is None:
            if anchors is None:
                raise ValueError("Without pretrained weights, `anchors` can't be empty.")
        else:
            pretrained_body = None
            if anchors is None:
                anchors = [[1, 1] for _ in range(self.pan_layers * self.abox_num)]
                use_arg_anchors = False
        if isinstance(pretrained_body, str):
            _offline("pretrained_body", pretrained_body)
        if backbone != "csp_darknet":
            if backbone in ("resnet50", "resnet101", "resnet152", "resnet50v2", "resnet101v2", "resnet152v2"):
                raise ValueError(f"backbone {backbone!r} lives in keras.applications, outside the HIP path")
            raise ValueError(f"Invalid backbone: {backbone}")
        model_body = bodies.yolo_body_v4(self.input_shape, pretrained_darknet=pretrained_body)
        self._model = bodies.yolo_head_v4(model_body, self.class_num, anchors, seed=seed,
                                          bn_unbiased_moving_var=bn_unbiased_moving_var)
        if pretrained_weights is not None:
            self._model.load_weights(pretrained_weights)
            if use_arg_anchors:
                self.anchors = anchors
                print("The saved model is loaded and will use the argument `anchors` instead of the original anchors.")
        self.grid_shape = tuple(self._model.output[0].shape[1:3])

    def loss(self, binary_weight=1, loss_weight=[1, 5, 1], wh_reg_weight=0.01, ignore_thresh=0.6, truth_thresh=1.0,
             label_smooth=0.0, focal_loss_gamma=2):
        if not isinstance(binary_weight, Iterable) or len(binary_weight) != self.pan_layers:
            binary_weight = [binary_weight] * self.pan_layers
        loss_weight = _loss_weight_list(loss_weight, ("box", "conf", "prob"))
        anchors = self.anchors
        loss_list = []
        for pan_id in range(self.pan_layers):
            amp = 2 ** pan_id
            grid_shape = (self.grid_shape[0] * amp, self.grid_shape[1] * amp)
            aid = self.abox_num * pan_id
            loss_list.append(losses.wrap_yolo_loss_v4(
                grid_shape=grid_shape, bbox_num=self.abox_num, class_num=self.class_num,
                anchors=anchors[aid:aid + self.abox_num], binary_weight=binary_weight[pan_id],
                loss_weight=loss_weight, wh_reg_weight=wh_reg_weight, ignore_thresh=ignore_thresh,
                truth_thresh=truth_thresh, label_smooth=label_smooth, focal_loss_gamma=focal_loss_gamma))
        return loss_list

    def metrics(self, kind="obj_acc"):
        out = []
        for pan_id in range(self.pan_layers):
            amp = 2 ** pan_id
            out.append(_metric_list(kind, 4, (self.grid_shape[0] * amp, self.grid_shape[1] * amp), self.abox_num,
                                    self.class_num))
        return out


# ---------------------------------------------------------------------------------------------
V2_DEFAULT_ANCHORS = [[0.75157846, 0.70525231], [0.60637077, 0.27136769], [0.25680231, 0.42110308],
                      [0.14418923, 0.15865615], [0.04405615, 0.05210654]]


class YoloV2(_IOUnavailable):
    def __init__(self, input_shape=(416, 416, 3), class_names=[]):
        self.input_shape = input_shape
        self.grid_shape = input_shape[0] // 32, input_shape[1] // 32
        self.abox_num = 5
        self.class_names = class_names
        self.class_num = len(class_names)
        self.anchors = None
        self.model = None
        self.file_names = None

    def create_model(self, anchors=V2_DEFAULT_ANCHORS, backbone="darknet", pretrained_weights=None,
                     pretrained_backbone=None, seed=1234, bn_unbiased_moving_var=True):
        if backbone != "darknet":
            if backbone in ("unet", "mobilenet"):
                raise ValueError(f"backbone {backbone!r} is outside the HIP path (SURVEY.md section 2 row 15)")
            raise ValueError(f"Invalid backbone: {backbone}")
        if isinstance(pretrained_backbone, str):
            _offline("pretrained_backbone", pretrained_backbone)
        model_body = bodies.yolo_body_v2(self.input_shape, backbone, pretrained_backbone)
        self.model = bodies.yolo_head_v2(model_body, self.class_num, anchors, seed=seed,
                                         bn_unbiased_moving_var=bn_unbiased_moving_var)
        if pretrained_weights is not None:
            self.model.load_weights(pretrained_weights)
        self.anchors = anchors
        self.abox_num = len(anchors)
        self.grid_shape = tuple(self.model.output.shape[1:3])

    def loss(self, binary_weight=1, loss_weight=[1, 1, 5, 1], ignore_thresh=0.6):
        loss_weight = _loss_weight_list(loss_weight, ("xy", "wh", "conf", "prob"))
        return losses.wrap_yolo_loss_v2(grid_shape=self.grid_shape, bbox_num=self.abox_num, class_num=self.class_num,
                                        anchors=self.anchors, binary_weight=binary_weight, loss_weight=loss_weight,
                                        ignore_thresh=ignore_thresh)

    def metrics(self, kind="obj_acc"):
        return _metric_list(kind, 2, self.grid_shape, self.abox_num, self.class_num)


# ---------------------------------------------------------------------------------------------
class YoloV1_5(_IOUnavailable):
    def __init__(self, input_shape=(448, 448, 3), class_names=[]):
        self.input_shape = input_shape
        self.grid_shape = input_shape[0] // 64, input_shape[1] // 64
        self.bbox_num = 2
        self.class_names = class_names
        self.class_num = len(class_names)
        self.model = None
        self.file_names = None

    def create_model(self, bbox_num=2, pretrained_weights=None, pretrained_backbone=None, seed=1234,
                     bn_unbiased_moving_var=True):
        model_body = bodies.yolo_body_v1(self.input_shape, pretrained_backbone)
        self.model = bodies.yolo_head_v1(model_body, bbox_num, self.class_num, seed=seed,
                                         bn_unbiased_moving_var=bn_unbiased_moving_var)
        if pretrained_weights is not None:
            self.model.load_weights(pretrained_weights)
        self.bbox_num = bbox_num
        self.grid_shape = tuple(self.model.output.shape[1:3])

    def loss(self, binary_weight, loss_weight=[5, 5, 1, 1]):
        loss_weight = _loss_weight_list(loss_weight, ("xy", "wh", "conf", "prob"))
        return losses.wrap_yolo_loss_v1(grid_shape=self.grid_shape, bbox_num=self.bbox_num, class_num=self.class_num,
                                        binary_weight=binary_weight, loss_weight=loss_weight)

    def metrics(self, kind="obj_acc"):
        return _metric_list(kind, 1, self.grid_shape, self.bbox_num, self.class_num)
