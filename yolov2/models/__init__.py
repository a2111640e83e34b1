"""Drop-in for the reference package `yolov2.models` (yolov2/models/__init__.py): yolo_body / yolo_head as graph-builder
entry points of the HIP executor (tf2_yolo_amd/bodies.py says what a body is here)."""
from tf2_yolo_amd.bodies import yolo_body_v2 as yolo_body, yolo_head_v2 as yolo_head  # noqa: F401
