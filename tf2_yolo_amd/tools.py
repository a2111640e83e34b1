"""Drop-in for the compute functions of the reference's utils/tools.py, running on the GPU.

  decode    utils/tools.py:370-438      nms       utils/tools.py:687-733
  soft_nms  utils/tools.py:736-786      (nms with iou_mode=2 is DIoU-NMS)

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


def _nms_impl(xywhcp, class_num, mode, nms_threshold, conf_threshold=0.5, sigma=0.5):
    was_numpy = not torch.is_tensor(xywhcp)
    rows = _to_dev(xywhcp)
    if rows.dtype != torch.float64:
        rows = rows.double()
    if rows.numel() == 0:
        rows = rows.reshape(0, 7)
    keep = ops.nms_keep(rows, class_num, mode, nms_threshold, conf_threshold, sigma).bool()
    # reference output order: classes ascending, original order inside a class (utils/tools.py:730-732)
    idx = torch.nonzero(keep).reshape(-1)
    cls = rows[idx, 5].to(torch.int64)
    order = torch.argsort(cls, stable=True)
    out = rows[idx[order]]
    return out.cpu().numpy() if was_numpy else out


def nms(xywhcp, class_num=1, nms_threshold=0.45, iou_mode=1):
    """Non-Maximum Suppression (iou_mode 1: IoU, 2: DIoU)."""
    if iou_mode not in (1, 2):
        raise ValueError(f"Invalid iou_mode: {iou_mode}")
    return _nms_impl(xywhcp, class_num, NMS_HARD if iou_mode == 1 else NMS_DIOU, nms_threshold)


def soft_nms(xywhcp, class_num=1, nms_threshold=0.45, conf_threshold=0.5, sigma=0.5):
    """Soft Non-Maximum Suppression."""
    return _nms_impl(xywhcp, class_num, NMS_SOFT, nms_threshold, conf_threshold, sigma)
