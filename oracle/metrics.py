"""Metric closures restated on torch CPU.

  v2/v3/v4: yolov3/metrics/yolo_metrics.py:9-115 (yolov2, yolov4 copies are identical)
  v1.5    : yolov1_5/metrics/yolo_metrics.py:9-107
`binary_accuracy(y_true, y_pred)` = mean(equal(y_true, cast(y_pred > 0.5)), axis=-1)
(SURVEY.md Appendix B).
"""
import torch

from .losses import cal_iou

EPSILON = 1e-07


def _split(y_true, y_pred, grid_shape, bbox_num, class_num):
    yt = y_true.reshape(-1, *grid_shape, 1, 5 + class_num).to(y_pred.dtype)
    yp = y_pred.reshape(-1, *grid_shape, bbox_num, 5 + class_num)
    return yt, yp


def obj_acc(y_true, y_pred, grid_shape, bbox_num, class_num):
    yt, yp = _split(y_true, y_pred, grid_shape, bbox_num, class_num)
    c_true = yt[..., 4]
    c_pred = yp[..., 4].max(dim=-1, keepdim=True).values
    return (c_true == (c_pred > 0.5).to(c_true.dtype)).to(c_true.dtype).mean(dim=-1)   # (N,gh,gw)


def mean_iou(y_true, y_pred, grid_shape, bbox_num, class_num):
    yt, yp = _split(y_true, y_pred, grid_shape, bbox_num, class_num)
    has_obj = yt[..., 4]
    iou = cal_iou(yt[..., :4], yp[..., :4], grid_shape).max(dim=-1, keepdim=True).values * has_obj
    return iou.sum() / (has_obj.sum() + EPSILON)


def class_acc(y_true, y_pred, grid_shape, bbox_num, class_num):
    yt, yp = _split(y_true, y_pred, grid_shape, bbox_num, class_num)
    has_obj = yt[..., 4]
    eq = (yt[..., -class_num:].argmax(-1) == yp[..., -class_num:].argmax(-1)).to(yt.dtype) * has_obj
    return eq.sum() / (has_obj.sum() * bbox_num + EPSILON)


def recall(y_true, y_pred, grid_shape, bbox_num, class_num, iou_threshold=0.5):
    yt, yp = _split(y_true, y_pred, grid_shape, bbox_num, class_num)
    has_obj = yt[..., 4]
    eq = (yt[..., -class_num:].argmax(-1) == yp[..., -class_num:].argmax(-1)).to(yt.dtype) * has_obj
    iou = (cal_iou(yt[..., :4], yp[..., :4], grid_shape) * eq).max(dim=-1, keepdim=True).values
    return (iou >= iou_threshold).to(yt.dtype).sum() / (has_obj.sum() + EPSILON)


# ---- v1.5 layout: [xywhc]*B + C ---------------------------------------------------------------
def _split_v1(y_true, y_pred, grid_shape, bbox_num, class_num):
    y_true = y_true.to(y_pred.dtype)
    t = y_true[..., :-class_num].reshape(-1, *grid_shape, 1, 5)
    p = y_pred[..., :-class_num].reshape(-1, *grid_shape, bbox_num, 5)
    return t, p


def obj_acc_v1(y_true, y_pred, grid_shape, bbox_num, class_num):
    t, p = _split_v1(y_true, y_pred, grid_shape, bbox_num, class_num)
    c_pred = p[..., 4].max(dim=-1, keepdim=True).values
    return (t[..., 4] == (c_pred > 0.5).to(t.dtype)).to(t.dtype).mean(dim=-1)


def mean_iou_v1(y_true, y_pred, grid_shape, bbox_num, class_num):
    t, p = _split_v1(y_true, y_pred, grid_shape, bbox_num, class_num)
    has_obj = t[..., 4]
    iou = cal_iou(t, p, grid_shape).max(dim=-1, keepdim=True).values * has_obj
    return iou.sum() / (has_obj.sum() + EPSILON)


def class_acc_v1(y_true, y_pred, grid_shape, class_num):
    y_true = y_true.to(y_pred.dtype)
    has_obj = y_true[..., :-class_num].reshape(-1, *grid_shape, 5)[..., 4]
    eq = (y_true[..., -class_num:].reshape(-1, *grid_shape, class_num).argmax(-1)
          == y_pred[..., -class_num:].reshape(-1, *grid_shape, class_num).argmax(-1)).to(y_true.dtype) * has_obj
    return eq.sum() / (has_obj.sum() + EPSILON)


def recall_v1(y_true, y_pred, grid_shape, bbox_num, class_num, iou_threshold=0.5):
    t, p = _split_v1(y_true, y_pred, grid_shape, bbox_num, class_num)
    y_true = y_true.to(y_pred.dtype)
    has_obj = t[..., 4]
    eq = (y_true[..., -class_num:].reshape(-1, *grid_shape, class_num).argmax(-1)
          == y_pred[..., -class_num:].reshape(-1, *grid_shape, class_num).argmax(-1)).to(y_true.dtype)
    eq = eq.unsqueeze(-1) * has_obj
    iou = (cal_iou(t, p, grid_shape) * eq).max(dim=-1, keepdim=True).values
    return (iou >= iou_threshold).to(y_true.dtype).sum() / (has_obj.sum() + EPSILON)
