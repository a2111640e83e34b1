"""Drop-in for the reference's utils/tools.py: the compute functions (decode, nms, soft_nms, cal_iou on the GPU;
down2xlabel, get_class_weight on the host, as in the reference), the detection writers (array_to_json,
array_to_xml), and explicit errors for the file readers / plotting that SURVEY.md section 2 rows 12-13 leave out of scope
(they need cv2 / imgaug / bs4 / matplotlib image I/O, absent in this environment)."""
from tf2_yolo_amd.labels import down2xlabel, get_class_weight  # noqa: F401  (utils/tools.py:342-367, :592-627)
from tf2_yolo_amd.tools import array_to_json, array_to_xml, cal_iou, decode, nms, soft_nms  # noqa: F401

EPSILON = 1e-07


def _io_unavailable(name):
    raise NotImplementedError(
        f"utils.tools.{name}: image / annotation file I/O and plotting are outside the accelerated hot path "
        "(cv2, imgaug, bs4 are not installed); feed ndarrays to model.fit / model.predict and use "
        "utils.tools.decode / nms for post-processing")


def read_img(path, size=(512, 512), rescale=None):
    """utils/tools.py:29-52 (cv2.imread + resize): out of scope."""
    _io_unavailable("read_img")


def vis_img(img, *label_datas, **kwargs):
    """utils/tools.py:441-589 (matplotlib drawing): out of scope."""
    _io_unavailable("vis_img")


class YoloDataSequence(object):
    """utils/tools.py:71-339 (annotation files -> augmented batches): out of scope; any object with
    __len__ / __getitem__(i) -> (images, labels) works as a Sequence for model.fit / model.predict."""

    def __init__(self, *args, **kwargs):
        _io_unavailable("YoloDataSequence")


def create_score_mat(*args, **kwargs):
    """Moved, as in the reference (utils/tools.py:789-797)."""
    raise ImportError("The location of this function has been changed. "
                      "Import it using"
                      "`from utils.measurement import create_score_mat`")
