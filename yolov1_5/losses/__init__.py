"""Drop-in for the reference package `yolov1_5.losses` (yolov1_5/losses/__init__.py:3): same names, HIP backend."""
from tf2_yolo_amd.losses import cal_iou_v3 as cal_iou  # noqa: F401  (yolov1_5/losses/loss.py:9-37)
from tf2_yolo_amd.losses import wrap_yolo_loss_v1 as wrap_yolo_loss  # noqa: F401
