"""Drop-in for the reference package `yolov1_5` (yolov1_5/__init__.py): same import surface, HIP backend."""
from tf2_yolo_amd.facade import MetricKind, YoloV1_5 as Yolo  # noqa: F401
from utils import tools  # noqa: F401
from .losses import wrap_yolo_loss  # noqa: F401
from .metrics import wrap_class_acc, wrap_mean_iou, wrap_obj_acc, wrap_recall  # noqa: F401
from .models import yolo_body, yolo_head  # noqa: F401
