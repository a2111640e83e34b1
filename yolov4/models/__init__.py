"""Drop-in for the reference package `yolov4.models` (yolov4/models/__init__.py): yolo_body / yolo_head as graph-builder
entry points of the HIP executor (tf2_yolo_amd/bodies.py says what a body is here)."""
from tf2_yolo_amd.bodies import csp_darknet53, yolo_keras_app_body  # noqa: F401
from tf2_yolo_amd.bodies import yolo_body_v4 as yolo_body, yolo_head_v4 as yolo_head  # noqa: F401
