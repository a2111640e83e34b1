"""Drop-in for the reference package `yolov3` (yolov3/__init__.py): same import surface, HIP backend."""
from tf2_yolo_amd.facade import MetricKind, YoloV3 as Yolo  # noqa: F401
from tf2_yolo_amd.losses import wrap_yolo_loss_v3 as wrap_yolo_loss  # noqa: F401
from tf2_yolo_amd.losses import wrap_class_acc, wrap_mean_iou, wrap_obj_acc, wrap_recall  # noqa: F401
