"""Loss / metric closures with the reference's factory signatures, backed by the fused HIP kernels.

  wrap_yolo_loss (v3)  yolov3/losses/loss.py:40-164      wrap_obj_acc / wrap_mean_iou /
  wrap_yolo_loss (v2)  yolov2/losses/loss.py:40-137      wrap_class_acc / wrap_recall:
  wrap_yolo_loss (v4)  yolov4/losses/loss.py:64-169        yolov3/metrics/yolo_metrics.py:9-115
  wrap_yolo_loss (v1)  yolov1_5/losses/loss.py:40-118      yolov1_5/metrics/yolo_metrics.py:9-107

A closure is called as f(y_true, y_pred) like a tf.keras loss and returns a 0-dim CUDA tensor.
y_true / y_pred may be NumPy arrays or tensors of any float dtype (cast to float32, as Keras
casts y_true to y_pred.dtype). `fwd_bwd` additionally returns dL/dy_pred from the same kernel
pass; the training loop (model.py) uses it instead of autodiff.
"""
import numpy as np
import torch

from . import ops


def _as_f32_cuda(a):
    if torch.is_tensor(a):
        return a.to(device="cuda", dtype=torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32))).cuda()


def _f32_cuda_view(a):
    if torch.is_tensor(a):
        return a.to(device="cuda", dtype=torch.float32)     # slices such as y[..., :4] stay views
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, dtype=np.float32))).cuda()


def cal_iou_v4(xywh_true, xywh_pred, grid_shape, return_ciou=False):
    """Calculate IOU of two tensors, loss-side form (yolov4/losses/loss.py:10-61): operands of shape (..., 4) that
    broadcast (the losses call it with (N,S,S,1,4) against (N,S,S,B,4)); x / y are cell offsets and are divided by
    grid_shape[::-1] first. float32 on the device like the reference's TF ops. Forward value only: inside the
    training step the fused loss kernel differentiates its own IoU / CIoU (csrc/loss.hip). Returns a CUDA tensor
    (two, IoU and CIoU, with return_ciou)."""
    a, b = _f32_cuda_view(xywh_true), _f32_cuda_view(xywh_pred)
    return ops.cal_iou(a, b, 3 if return_ciou else 1, grid_wh=(grid_shape[1], grid_shape[0]))


def cal_iou_v3(xywh_true, xywh_pred, grid_shape):
    """yolov3/losses/loss.py:9-37 (the same text in yolov1_5 / yolov2): IoU only."""
    return cal_iou_v4(xywh_true, xywh_pred, grid_shape)


class YoloLoss:
    def __init__(self, version, grid_shape, bbox_num, class_num, anchors=None, **kw):
        self.version = version
        self.grid_shape = (int(grid_shape[0]), int(grid_shape[1]))
        self.bbox_num, self.class_num = int(bbox_num), int(class_num)
        self.anchors = None if anchors is None else [tuple(map(float, a)) for a in np.asarray(anchors).reshape(-1, 2)]
        self.kw = kw
        self.__name__ = "yolo_loss"

    def cfg(self, N):
        return ops.make_loss_cfg(self.version, N, self.grid_shape[0], self.grid_shape[1], self.bbox_num,
                                 self.class_num, self.anchors, **self.kw)

    def _prep(self, y_true, y_pred):
        yt, yp = _as_f32_cuda(y_true), _as_f32_cuda(y_pred)
        cells = self.grid_shape[0] * self.grid_shape[1]
        N = yt.numel() // (cells * (5 + self.class_num))
        return yt, yp, N

    def fwd_bwd(self, y_true, y_pred, grad_scale=1.0, dpred=None, loss_out=None, decisions=None):
        yt, yp, N = self._prep(y_true, y_pred)
        return ops.loss_fwd_bwd(self.cfg(N), yt, yp, loss_out=loss_out, dpred=dpred, grad_scale=grad_scale,
                                decisions=decisions)

    def parts(self, y_true, y_pred):
        yt, yp, N = self._prep(y_true, y_pred)
        out, _ = ops.loss_fwd_bwd(self.cfg(N), yt, yp, want_grad=False)
        return out

    def __call__(self, y_true, y_pred):
        return self.parts(y_true, y_pred)[0].float()


def wrap_yolo_loss_v3(grid_shape, bbox_num, class_num, anchors=None, binary_weight=1, loss_weight=[1, 1, 1, 1],
                      ignore_thresh=.6, use_focal_loss=False, focal_loss_gamma=2, use_scale=True):
    return YoloLoss(3, grid_shape, bbox_num, class_num, anchors, binary_weight=binary_weight,
                    loss_weight=loss_weight, ignore_thresh=ignore_thresh, use_focal_loss=use_focal_loss,
                    focal_gamma=focal_loss_gamma, use_scale=use_scale)


def wrap_yolo_loss_v2(grid_shape, bbox_num, class_num, anchors, binary_weight=1, loss_weight=[1, 1, 1, 1],
                      ignore_thresh=.6):
    return YoloLoss(2, grid_shape, bbox_num, class_num, anchors, binary_weight=binary_weight,
                    loss_weight=loss_weight, ignore_thresh=ignore_thresh)


def wrap_yolo_loss_v4(grid_shape, bbox_num, class_num, anchors=None, binary_weight=1, loss_weight=[1, 1, 1],
                      wh_reg_weight=0.01, ignore_thresh=.6, truth_thresh=1, label_smooth=0, focal_loss_gamma=2):
    return YoloLoss(4, grid_shape, bbox_num, class_num, anchors, binary_weight=binary_weight,
                    loss_weight=loss_weight, wh_reg_weight=wh_reg_weight, ignore_thresh=ignore_thresh,
                    truth_thresh=truth_thresh, label_smooth=label_smooth, focal_gamma=focal_loss_gamma)


def wrap_yolo_loss_v1(grid_shape, bbox_num, class_num, binary_weight=1, loss_weight=[1, 1, 1, 1]):
    return YoloLoss(1, grid_shape, bbox_num, class_num, None, binary_weight=binary_weight, loss_weight=loss_weight)


# ---- metrics -------------------------------------------------------------------------------------
class YoloMetric:
    """kind in {obj_acc, mean_iou, class_acc, recall}; one kernel pass yields all numerators, the
    closure picks its own ratio. obj_acc returns the scalar mean (Keras means the (N,gh,gw) tensor
    the reference returns)."""

    def __init__(self, kind, version, grid_shape, bbox_num, class_num, iou_threshold=0.5):
        self.kind, self.version = kind, version
        self.grid_shape = (int(grid_shape[0]), int(grid_shape[1]))
        self.bbox_num, self.class_num, self.iou_threshold = int(bbox_num), int(class_num), float(iou_threshold)
        self.__name__ = kind if kind != "recall" else "recall"

    def raw(self, y_true, y_pred):
        yt, yp = _as_f32_cuda(y_true), _as_f32_cuda(y_pred)
        cells = self.grid_shape[0] * self.grid_shape[1]
        N = yt.numel() // (cells * (5 + self.class_num))
        cfg = ops.make_loss_cfg(self.version, N, self.grid_shape[0], self.grid_shape[1], self.bbox_num,
                                self.class_num, None)
        return ops.metrics(cfg, yt, yp, self.iou_threshold)

    def from_raw(self, o):
        eps = 1e-07
        if self.kind == "obj_acc":
            return (o[0] / o[5]).float()
        if self.kind == "mean_iou":
            return (o[1] / (o[2] + eps)).float()
        if self.kind == "class_acc":
            denom = o[2] * (1 if self.version == 1 else self.bbox_num) + eps
            return (o[3] / denom).float()
        return (o[4] / (o[2] + eps)).float()

    def __call__(self, y_true, y_pred):
        return self.from_raw(self.raw(y_true, y_pred))


def wrap_obj_acc(grid_shape, bbox_num, class_num, version=3):
    return YoloMetric("obj_acc", version, grid_shape, bbox_num, class_num)


def wrap_mean_iou(grid_shape, bbox_num, class_num, version=3):
    return YoloMetric("mean_iou", version, grid_shape, bbox_num, class_num)


def wrap_class_acc(grid_shape, bbox_num, class_num, version=3):
    return YoloMetric("class_acc", version, grid_shape, bbox_num, class_num)


def wrap_recall(grid_shape, bbox_num, class_num, iou_threshold=0.5, version=3):
    return YoloMetric("recall", version, grid_shape, bbox_num, class_num, iou_threshold)
