"""Drop-in for the compute functions of the reference's utils/tools.py, running on the GPU.

  decode    utils/tools.py:370-438      nms       utils/tools.py:687-733
  soft_nms  utils/tools.py:736-786      (nms with iou_mode=2 is DIoU-NMS)
  cal_iou   utils/tools.py:630-684      (broadcasting IoU / DIoU matrix)

Same names, argument meaning, return types (NumPy float64 arrays of shape (n, 7)) and error
behaviour (ValueError("Invalid version: ...")). Inputs may be NumPy arrays (uploaded) or CUDA
tensors (used in place). Index selection is bit-exact with the reference; the only documented
difference is the order of exact score ties, which the reference leaves undefined
(np.argsort is unstable) and which is "higher original index first" here.
"""
import numpy as np
import torch

from . import ops
from ._lib import NMS_DIOU, NMS_HARD, NMS_SOFT, YoloHipError


def _device():
    if not torch.cuda.is_available():
        raise YoloHipError("tf2_yolo_amd.tools needs a HIP device: there is no CPU fallback")
    return torch.device("cuda")


def _to_dev(a):
    if torch.is_tensor(a):
        if not a.is_cuda:
            a = a.cuda()
        if a.dtype not in (torch.float32, torch.float64):
            a = a.double()
        return a.contiguous()
    a = np.asarray(a)
    if a.dtype not in (np.float32, np.float64):
        a = a.astype(np.float64)
    return torch.from_numpy(np.ascontiguousarray(a)).to(_device())


def decode_device(*label_datas, class_num=1, threshold=0.5, version=1, capacity=None):
    """GPU-resident decode: returns a float64 CUDA tensor (n, 7)."""
    if version not in (1, 2, 3, 4):
        raise ValueError(f"Invalid version: {version}")
    dev = _device()
    levels = [_to_dev(a) for a in label_datas]
    geoms = []
    total_slots = 0
    for lv in levels:
        if lv.dim() != 3:
            raise ValueError("each label_data must have shape (grid_heights, grid_widths, info)")
        gh, gw, d = lv.shape
        A = (d - class_num) // 5 if version == 1 else d // (5 + class_num)
        geoms.append((gh, gw, A))
        total_slots += gh * gw * A * class_num
    if capacity is None:
        capacity = min(total_slots, 1 << 16)
    capacity = max(int(capacity), 1)
    count = torch.zeros(1, device=dev, dtype=torch.int32)
    while True:
        rows = torch.empty((capacity, 7), device=dev, dtype=torch.float64)
        count.zero_()
        for lv, (gh, gw, A) in zip(levels, geoms):
            ws = torch.empty(max(ops.decode_workspace_bytes(gh, gw, A, class_num), 4), device=dev, dtype=torch.uint8)
            thr = float(np.float32(threshold)) if lv.dtype == torch.float32 else float(threshold)
            ops.decode_level(lv, A, class_num, version, thr, rows, capacity, count, ws)
        n = int(count.item())
        if n <= capacity:
            return rows[:n]
        capacity = n   # rare: more candidates than the first guess; rerun with the exact size


def decode(*label_datas, class_num=1, threshold=0.5, version=1):
    """Decode the prediction from yolo model -> ndarray (N, 7): x, y, w, h, c, class index, class prob."""
    out = decode_device(*label_datas, class_num=class_num, threshold=threshold, version=version)
    if out.shape[0] == 0:
        return np.array([], dtype="float")     # what np.array([]) of an empty list gives in the reference
    return out.cpu().numpy()


def cal_iou(xywh_true, xywh_pred, mode=1):
    """Calculate IOU of two tensors (utils/tools.py:630-684): arrays of shape (..., >=4) that broadcast against each
    other, (x, y) normalised by the image size; mode 1: IoU, 2: DIoU. Computed on the GPU in the operands' NumPy
    result type (float64 unless both are float32) with the reference's operation order -- bit-identical to the
    reference's arrays. NumPy in -> NumPy out, CUDA tensors in -> CUDA tensor out."""
    if mode not in (1, 2):
        return None   # the reference falls through both branches and returns None
    was_numpy = not (torch.is_tensor(xywh_true) or torch.is_tensor(xywh_pred))
    a, b = _to_dev(xywh_true), _to_dev(xywh_pred)
    if a.dtype != b.dtype:
        a, b = a.double(), b.double()
    out = ops.cal_iou(a, b, mode)
    return out.cpu().numpy() if was_numpy else out


def _nms_impl(xywhcp, class_num, mode, nms_threshold, conf_threshold=0.5, sigma=0.5):
    was_numpy = not torch.is_tensor(xywhcp)
    rows = _to_dev(xywhcp)
    if rows.dtype != torch.float64:
        rows = rows.double()
    if rows.numel() == 0:
        rows = rows.reshape(0, 7)
    # NMS + the reference's output order (classes ascending, original order inside a class: utils/tools.py:730-732) on the
    # device: the kept rows arrive gathered
    out = ops.nms_select(rows.contiguous(), class_num, mode, nms_threshold, conf_threshold, sigma)
    return out.cpu().numpy() if was_numpy else out


def nms(xywhcp, class_num=1, nms_threshold=0.45, iou_mode=1):
    """Non-Maximum Suppression (iou_mode 1: IoU, 2: DIoU)."""
    if iou_mode not in (1, 2):
        raise ValueError(f"Invalid iou_mode: {iou_mode}")
    return _nms_impl(xywhcp, class_num, NMS_HARD if iou_mode == 1 else NMS_DIOU, nms_threshold)


def soft_nms(xywhcp, class_num=1, nms_threshold=0.45, conf_threshold=0.5, sigma=0.5):
    """Soft Non-Maximum Suppression."""
    return _nms_impl(xywhcp, class_num, NMS_SOFT, nms_threshold, conf_threshold, sigma)


# ---- on-disk formats of detections (utils/tools.py:800-965) ---------------------------------------------
def _detections(label_datas, class_num, conf_threshold, nms_mode, nms_threshold, nms_sigma, version):
    """decode (+ NMS mode 1: NMS, 2: soft-NMS, 3: DIoU-NMS) on the GPU -> NumPy rows (n, 7)"""
    rows = decode_device(*label_datas, class_num=class_num, threshold=conf_threshold, version=version)
    if nms_mode > 0 and rows.shape[0] > 0:
        if nms_mode == 1:
            rows = nms(rows, class_num, nms_threshold)
        elif nms_mode == 2:
            rows = soft_nms(rows, class_num, nms_threshold, conf_threshold, nms_sigma)
        elif nms_mode == 3:
            rows = nms(rows, class_num, nms_threshold, 2)
    return rows.cpu().numpy().reshape(-1, 7)


def _pixel_boxes(rows, img_size):
    """(label index, x_min, y_min, x_max, y_max, confidence) per row, in pixels of the original image"""
    out = []
    for obj in rows:
        box_x, box_y = obj[0] * img_size[1], obj[1] * img_size[0]
        box_w, box_h = obj[2] * img_size[1], obj[3] * img_size[0]
        out.append((int(obj[5]), box_x - box_w / 2, box_y - box_h / 2, box_x + box_w / 2, box_y + box_h / 2,
                    obj[4] * obj[6]))
    return out


def array_to_json(path, img_size, *label_datas, class_names=[""], conf_threshold=0.5, nms_mode=0,
                  nms_threshold=0.45, nms_sigma=0.5, version=3):
    """Write the detections of one image as a labelme-style json file (utils/tools.py:800-876): rectangles as
    [[x_min, y_min], [x_max, y_max]] in pixels plus "confidence" = conf * class prob, encoding big5.
    Numbers are written as plain decimals (what the reference writes under NumPy 1.x; under NumPy 2 its
    str(dict) leaks `np.float64(...)` wrappers into the file, which is not json)."""
    rows = _detections(label_datas, len(class_names), conf_threshold, nms_mode, nms_threshold, nms_sigma, version)
    shapes = []
    for ci, x0, y0, x1, y1, conf in _pixel_boxes(rows, img_size):
        shapes.append('{"label": "%s", "points": [[%r, %r], [%r, %r]], "shape_type": "rectangle", "confidence": %r}'
                      % (class_names[ci], float(x0), float(y0), float(x1), float(y1), float(conf)))
    text = '{"shapes": [%s], "imageHeight": %r, "imageWidth": %r}' % (", ".join(shapes), img_size[0], img_size[1])
    with open(path, "w", encoding="big5") as file:
        file.write(text)


def array_to_xml(path, img_size, *label_datas, class_names=[], conf_threshold=0.5, nms_mode=0,
                 nms_threshold=0.45, nms_sigma=0.5, version=3):
    """Write the detections of one image as a VOC-style xml file (utils/tools.py:879-965): per object <name>,
    <bndbox> with integer (truncated) xmin / ymin / xmax / ymax in pixels, <confidence>."""
    import xml.etree.ElementTree as ET
    rows = _detections(label_datas, len(class_names), conf_threshold, nms_mode, nms_threshold, nms_sigma, version)
    root = ET.Element("annotation")
    for ci, x0, y0, x1, y1, conf in _pixel_boxes(rows, img_size):
        node = ET.SubElement(root, "object")
        ET.SubElement(node, "name").text = class_names[ci]
        box = ET.SubElement(node, "bndbox")
        for tag, v in (("xmin", x0), ("ymin", y0), ("xmax", x1), ("ymax", y1)):
            ET.SubElement(box, tag).text = str(int(v))
        ET.SubElement(node, "confidence").text = str(float(conf))
    with open(path, "wb") as file:
        ET.ElementTree(root).write(file)
