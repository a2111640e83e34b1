"""Drop-in for the compute functions of the reference's utils/tools.py (decode, nms, soft_nms) and its
detection writers (array_to_json, array_to_xml)."""
from tf2_yolo_amd.tools import array_to_json, array_to_xml, decode, nms, soft_nms  # noqa: F401


def create_score_mat(*args, **kwargs):
    """Moved, as in the reference (utils/tools.py:789-797)."""
    raise ImportError("The location of this function has been changed. "
                      "Import it using"
                      "`from utils.measurement import create_score_mat`")
