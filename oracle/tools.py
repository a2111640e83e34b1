"""NumPy restatement of the reference's post-processing and label utilities.

  decode           utils/tools.py:370-438
  cal_iou          utils/tools.py:630-684
  nms              utils/tools.py:687-733
  soft_nms         utils/tools.py:736-786
  down2xlabel      utils/tools.py:342-367
  get_class_weight utils/tools.py:592-627
  encode_boxes     utils/tools.py:179-209  (box -> grid label, last writer wins per cell)

Pinned against the reference's own outputs in tests/golden/*.npz (tests/test_oracle_golden.py).
The loops of the reference are replaced by equivalent array code where the result is provably
the same (documented per function); the greedy NMS walk is kept sequential.
"""
import numpy as np

EPSILON = 1e-07


def decode(*label_datas, class_num=1, threshold=0.5, version=1):
    """Rows (x, y, w, h, conf, class, prob), float64, in np.where (C) order per level."""
    output = []
    for label_data in label_datas:
        grid_shape = label_data.shape[:2]
        if version == 1:
            bbox_num = (label_data.shape[-1] - class_num) // 5
            xywhc = np.reshape(label_data[..., :-class_num], (*grid_shape, bbox_num, 5))
            prob = np.expand_dims(label_data[..., -class_num:], axis=-2)
        elif version in (2, 3, 4):
            bbox_num = label_data.shape[-1] // (5 + class_num)
            label_data = np.reshape(label_data, (*grid_shape, bbox_num, 5 + class_num))
            xywhc = label_data[..., :5]
            prob = label_data[..., -class_num:]
        else:
            raise ValueError(f"Invalid version: {version}")
        joint_conf = xywhc[..., 4:5] * prob            # product in the input dtype (fp32 for predictions)
        y_i, x_i, box_i, class_i = np.where(joint_conf >= threshold)
        if len(y_i) == 0:
            continue
        x_reg = xywhc[y_i, x_i, box_i, 0]
        y_reg = xywhc[y_i, x_i, box_i, 1]
        rows = np.empty((len(y_i), 7), dtype=np.float64)
        rows[:, 0] = (x_i + x_reg) / grid_shape[1]     # int64 + float32 -> float64 (NumPy promotion)
        rows[:, 1] = (y_i + y_reg) / grid_shape[0]
        rows[:, 2] = xywhc[y_i, x_i, box_i, 2]
        rows[:, 3] = xywhc[y_i, x_i, box_i, 3]
        rows[:, 4] = xywhc[y_i, x_i, box_i, 4]
        rows[:, 5] = class_i
        rows[:, 6] = prob[y_i, x_i, 0, class_i] if version == 1 else prob[y_i, x_i, box_i, class_i]
        output.append(rows)
    if not output:
        return np.array([], dtype="float")
    return np.vstack(output)


def cal_iou(xywh_true, xywh_pred, mode=1):
    xy_true, wh_true = xywh_true[..., 0:2], xywh_true[..., 2:4]
    xy_pred, wh_pred = xywh_pred[..., 0:2], xywh_pred[..., 2:4]
    half_wh_true = wh_true / 2.
    mins_true, maxes_true = xy_true - half_wh_true, xy_true + half_wh_true
    half_wh_pred = wh_pred / 2.
    mins_pred, maxes_pred = xy_pred - half_wh_pred, xy_pred + half_wh_pred
    intersect_mins = np.maximum(mins_pred, mins_true)
    intersect_maxes = np.minimum(maxes_pred, maxes_true)
    intersect_wh = np.maximum(intersect_maxes - intersect_mins, 0.)
    intersect_areas = intersect_wh[..., 0] * intersect_wh[..., 1]
    true_areas = wh_true[..., 0] * wh_true[..., 1]
    pred_areas = wh_pred[..., 0] * wh_pred[..., 1]
    union_areas = pred_areas + true_areas - intersect_areas
    iou_scores = intersect_areas / (union_areas + EPSILON)
    if mode == 1:
        return iou_scores
    enclose_mins = np.minimum(mins_pred, mins_true)
    enclose_maxes = np.maximum(maxes_pred, maxes_true)
    enclose_wh = enclose_maxes - enclose_mins
    enclose_c2 = np.power(enclose_wh[..., 0], 2) + np.power(enclose_wh[..., 1], 2)
    p_rho2 = (np.power(xy_true[..., 0] - xy_pred[..., 0], 2) + np.power(xy_true[..., 1] - xy_pred[..., 1], 2))
    with np.errstate(invalid="ignore", divide="ignore"):
        return iou_scores - p_rho2 / enclose_c2


def _sorted_desc(conf):
    """The reference uses np.argsort(conf)[::-1] (unstable on ties). Ties are defined here as
    "higher original index first" = reversed STABLE ascending sort."""
    return np.argsort(conf, kind="stable")[::-1]


def nms_keep(xywhcp, class_num=1, nms_threshold=0.45, iou_mode=1):
    """Boolean keep mask over the input rows (rows of classes outside [0,class_num) are dropped,
    as the reference's per-class gather does)."""
    n = len(xywhcp)
    keep = np.zeros(n, dtype=bool)
    if n == 0:
        return keep
    cls = xywhcp[..., 5].astype("int")
    for i_class in range(class_num):
        idx = np.nonzero(cls == i_class)[0]
        if len(idx) == 0:
            continue
        rows = xywhcp[idx]
        box = rows[..., :5]
        iou = cal_iou(box.reshape(-1, 1, 5), box.reshape(1, -1, 5), mode=iou_mode)
        conf = box[..., 4] * rows[..., 6]
        order = _sorted_desc(conf)
        visited = np.zeros(len(idx), dtype=bool)
        deleted = np.zeros(len(idx), dtype=bool)
        for ci in order:
            visited[ci] = True
            if not deleted[ci]:
                with np.errstate(invalid="ignore"):
                    hit = iou[ci] >= nms_threshold
                deleted |= hit & ~visited
        keep[idx[~deleted]] = True
    return keep


def soft_nms_keep(xywhcp, class_num=1, nms_threshold=0.45, conf_threshold=0.5, sigma=0.5):
    n = len(xywhcp)
    keep = np.zeros(n, dtype=bool)
    if n == 0:
        return keep
    cls = xywhcp[..., 5].astype("int")
    for i_class in range(class_num):
        idx = np.nonzero(cls == i_class)[0]
        if len(idx) == 0:
            continue
        rows = xywhcp[idx]
        box = rows[..., :5]
        iou = cal_iou(box.reshape(-1, 1, 5), box.reshape(1, -1, 5))
        conf = box[..., 4] * rows[..., 6]
        order = _sorted_desc(conf)
        visited = np.zeros(len(idx), dtype=bool)
        deleted = np.zeros(len(idx), dtype=bool)
        for ci in order:
            visited[ci] = True
            for oi in np.where(iou[ci] >= nms_threshold)[0]:
                if not visited[oi]:
                    conf[oi] *= np.exp(-1 * (iou[ci][oi] ** 2) / sigma)
                    if conf[oi] < conf_threshold:
                        deleted[oi] = True
        keep[idx[~deleted]] = True
    return keep


def _gather_by_class(xywhcp, keep, class_num):
    cls = xywhcp[..., 5].astype("int")
    parts = [xywhcp[(cls == c) & keep] for c in range(class_num)]
    return np.vstack(parts) if parts else xywhcp[:0]


def nms(xywhcp, class_num=1, nms_threshold=0.45, iou_mode=1):
    return _gather_by_class(xywhcp, nms_keep(xywhcp, class_num, nms_threshold, iou_mode), class_num)


def soft_nms(xywhcp, class_num=1, nms_threshold=0.45, conf_threshold=0.5, sigma=0.5):
    return _gather_by_class(xywhcp, soft_nms_keep(xywhcp, class_num, nms_threshold, conf_threshold, sigma),
                            class_num)


def down2xlabel(label_data):
    batches, grid_h, grid_w, channels = label_data.shape
    new_label = np.zeros((batches, grid_h // 2, grid_w // 2, channels))
    for batch in range(batches):
        for i in range(0, grid_h, 2):
            for j in range(0, grid_w, 2):
                crop = label_data[batch][i:i + 2, j:j + 2]
                if crop[..., 4].max() == 1:
                    max_id = (crop[..., 2] * crop[..., 3]).argmax()
                    crop = crop[max_id // 2, max_id % 2]
                    new_label[batch][i // 2, j // 2, :2] = (crop[:2] + [max_id % 2, max_id // 2]) / 2
                    new_label[batch][i // 2, j // 2, 2:] = crop[2:]
    return new_label


def get_class_weight(label_data, method="alpha"):
    class_weight = []
    if method != "alpha":
        total = 1
        for i in label_data.shape[:-1]:
            total *= i
        if method == "effective":
            beta = (total - 1) / total
    for i in range(label_data.shape[-1]):
        samples_per_class = label_data[..., i].sum()
        if method == "effective":
            class_weight.append((1 - beta) / (1 - np.power(beta, samples_per_class)))
        elif method == "binary":
            class_weight.append(samples_per_class / (total - samples_per_class))
        else:
            class_weight.append(1 / samples_per_class)
    class_weight = np.array(class_weight)
    if method == "log":
        class_weight = np.log(total * class_weight)
    if method != "binary":
        class_weight = class_weight / np.sum(class_weight) * len(class_weight)
    return class_weight


def encode_boxes(boxes, labels, img_hw, grid_shape, class_num):
    """One image's label tensor from pixel boxes (x1,y1,x2,y2): utils/tools.py:179-209."""
    label = np.zeros((grid_shape[0], grid_shape[1], 5 + class_num))
    img_h, img_w = img_hw
    cell_h, cell_w = img_h / grid_shape[0], img_w / grid_shape[1]
    for (x1, y1, x2, y2), lab in zip(boxes, labels):
        bx, by, bw, bh = x1 + (x2 - x1) / 2, y1 + (y2 - y1) / 2, x2 - x1, y2 - y1
        x_i, y_i = int(bx // cell_w), int(by // cell_h)
        if x_i < grid_shape[1] and y_i < grid_shape[0]:
            label[y_i, x_i, 0] = bx % cell_w / cell_w
            label[y_i, x_i, 1] = by % cell_h / cell_h
            label[y_i, x_i, 2] = bw / img_w
            label[y_i, x_i, 3] = bh / img_h
            label[y_i, x_i, 4] = 1
            label[y_i, x_i, 5 + lab] = 1
    return label
