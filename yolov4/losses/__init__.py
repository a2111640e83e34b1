"""Drop-in for the reference package `yolov4.losses` (yolov4/losses/__init__.py:3): same names, HIP backend."""
from tf2_yolo_amd.losses import cal_iou_v4 as cal_iou  # noqa: F401  (yolov4/losses/loss.py:9-61)
from tf2_yolo_amd.losses import wrap_yolo_loss_v4 as wrap_yolo_loss  # noqa: F401
