"""Drop-in for the compute functions of the reference's utils/tools.py (decode, nms, soft_nms)."""
from tf2_yolo_amd.tools import decode, nms, soft_nms  # noqa: F401
