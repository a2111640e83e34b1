"""Drop-in for the reference package `yolov1_5.metrics` (yolov1_5/metrics/yolo_metrics.py): the four metric factories with
the reference's argument lists; each returns f(y_true, y_pred) -> 0-dim CUDA tensor (csrc/loss.hip:metrics_kernel)."""
from tf2_yolo_amd import losses as _l


def wrap_obj_acc(grid_shape, bbox_num, class_num):
    return _l.wrap_obj_acc(grid_shape, bbox_num, class_num, version=1)


def wrap_mean_iou(grid_shape, bbox_num, class_num):
    return _l.wrap_mean_iou(grid_shape, bbox_num, class_num, version=1)


def wrap_class_acc(grid_shape, class_num):
    """two arguments in v1.5 (yolov1_5/metrics/yolo_metrics.py:52): the class probabilities are per cell"""
    return _l.wrap_class_acc(grid_shape, 1, class_num, version=1)


def wrap_recall(grid_shape, bbox_num, class_num, iou_threshold=0.5):
    return _l.wrap_recall(grid_shape, bbox_num, class_num, iou_threshold=iou_threshold, version=1)
