"""Drop-in for the reference package `yolov3.models` (yolov3/models/__init__.py): yolo_body / yolo_head as graph-builder
entry points of the HIP executor (tf2_yolo_amd/bodies.py says what a body is here)."""
from tf2_yolo_amd.bodies import darknet53, tiny_yolo_body, yolo_keras_app_body  # noqa: F401
from tf2_yolo_amd.bodies import yolo_body_v3 as yolo_body, yolo_head_v3 as yolo_head  # noqa: F401
