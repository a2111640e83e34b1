"""Drop-in for the reference package `yolov1_5.models` (yolov1_5/models/__init__.py): yolo_body / yolo_head as graph-builder
entry points of the HIP executor (tf2_yolo_amd/bodies.py says what a body is here)."""
from tf2_yolo_amd.bodies import darknet  # noqa: F401
from tf2_yolo_amd.bodies import yolo_body_v1 as yolo_body, yolo_head_v1 as yolo_head  # noqa: F401
