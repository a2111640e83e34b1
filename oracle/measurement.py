"""CPU restatement (NumPy) of the reference's detection evaluation -- TEST INFRASTRUCTURE ONLY.

Follows utils/measurement.py of the reference: create_score_mat (:16-150) and PRfunc.__init__ (:198-326),
__call__ (:328-338), get_map (:393-447), on top of oracle/tools.py (decode, nms, soft_nms, cal_iou).
PINNED: checked bit for bit against outputs of the reference's own utils/measurement.py, executed in the
build container by tests/golden/make_measurement_golden.py (tests/golden/measurement_golden.npz).
Only tests/ may import this module.
"""
import numpy as np

from . import tools as T


def _rows(y_true, y_pred, class_num, conf_threshold, nms_mode, nms_threshold, nms_sigma, version):
    """measurement.py:77-92 / :216-233: decoded ground truth rows, decoded (+NMS) detection rows"""
    gt = np.asarray(T.decode(y_true, class_num=class_num, version=version), dtype=np.float64).reshape(-1, 7)
    det = np.asarray(T.decode(*y_pred, class_num=class_num, threshold=conf_threshold, version=version),
                     dtype=np.float64).reshape(-1, 7)
    if nms_mode > 0 and len(det) > 0:
        if nms_mode == 1:
            det = T.nms(det, class_num, nms_threshold)
        elif nms_mode == 2:
            det = T.soft_nms(det, class_num, nms_threshold, conf_threshold, nms_sigma)
        elif nms_mode == 3:
            det = T.nms(det, class_num, nms_threshold, 2)
    return gt, np.asarray(det, dtype=np.float64).reshape(-1, 7)


def _match(gt_c, det_c, iou_threshold):
    """measurement.py:116-128: best ground truth (first maximum) and hit flag per detection of one class"""
    iou = T.cal_iou(gt_c[:, :5].reshape(-1, 1, 5), det_c[:, :5].reshape(1, -1, 5))
    return np.argmax(iou, axis=0), np.max(iou, axis=0) >= iou_threshold


def score_counts(y_trues, y_preds, class_num, conf_threshold=0.5, nms_mode=0, nms_threshold=0.5, nms_sigma=0.5,
                 iou_threshold=0.5, version=3):
    """per class [detections, ground truths, matched detections, distinct matched ground truths]"""
    counts = np.zeros((class_num, 4), dtype=np.int64)
    for i in range(len(y_trues)):
        gt, det = _rows(y_trues[i], [p[i] for p in y_preds], class_num, conf_threshold, nms_mode, nms_threshold,
                        nms_sigma, version)
        gcls, dcls = gt[:, 5].astype("int"), det[:, 5].astype("int")
        for c in range(class_num):
            g, d = gt[gcls == c], det[dcls == c]
            counts[c, 0] += len(d)
            counts[c, 1] += len(g)
            if len(g) and len(d):
                arg, hit = _match(g, d, iou_threshold)
                counts[c, 2] += int(hit.sum())
                counts[c, 3] += len(set(arg[hit]))
    return counts


def score_table(counts, precision_mode=2):
    """measurement.py:130-143: precision, recall, F1 arrays from the accumulated counts"""
    c = counts.astype(np.float64)
    dets, gts, tpp, tp = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
    with np.errstate(divide="ignore", invalid="ignore"):
        precision = (tpp if precision_mode == 0 else tp) / (dets - (tpp - tp) if precision_mode == 1 else dets)
        recall = tp / gts
        f1 = (2 * precision * recall) / (precision + recall)
    return precision, recall, f1


def pr_curves(y_trues, y_preds, class_num, conf_threshold=0.05, nms_mode=1, nms_threshold=0.5, nms_sigma=0.5,
              iou_threshold=0.5, precision_mode=2, max_per_img=100, version=3):
    """measurement.py:209-323 -> (precisions, recalls): per class arrays of n+1 points"""
    gts = [0] * class_num
    dets = [np.empty((0, 3)) for _ in range(class_num)]
    for i in range(len(y_trues)):
        gt, det = _rows(y_trues[i], [p[i] for p in y_preds], class_num, conf_threshold, nms_mode, nms_threshold,
                        nms_sigma, version)
        gcls, dcls = gt[:, 5].astype("int"), det[:, 5].astype("int")
        for c in range(class_num):
            g, d = gt[gcls == c], det[dcls == c]
            seen = gts[c]
            gts[c] = seen + len(g)
            if len(d) == 0:
                continue
            joint = d[:, 4] * d[:, 6]
            if len(g):
                arg, hit = _match(g, d, iou_threshold)
                rec = np.stack((joint, arg + seen, hit.astype(np.float64)), axis=1)
            else:
                rec = np.stack((joint, np.zeros(len(d)), np.zeros(len(d))), axis=1)
            if max_per_img is not None and len(rec) > max_per_img:
                rec = rec[np.argsort(rec[:, 0])[::-1]][:max_per_img]
            dets[c] = np.vstack((dets[c], rec))
    precisions, recalls = [], []
    for c in range(class_num):
        rec = dets[c][np.argsort(dets[c][:, 0])[::-1]]
        n = len(rec)
        hit = rec[:, 2].astype(bool)
        ps, rs = np.zeros(n + 1), np.zeros(n + 1)
        seen_ids, tpp = set(), 0
        for k in range(n):   # prefix statistics; the reference recomputes them from scratch for every k
            if hit[k]:
                tpp += 1
                seen_ids.add(rec[k, 1])
            tp, nd = len(seen_ids), k + 1
            ps[k] = (tpp / nd) if precision_mode == 0 else (tp / (tp + nd - tpp)) if precision_mode == 1 else (tp / nd)
            rs[k] = tp / gts[c]
        rs[n] = rs[n - 1] if n else (0.0 if gts[c] else np.nan)
        precisions.append(ps)
        recalls.append(rs)
    return precisions, recalls


def precision_at(precisions, recalls, recall):
    """measurement.py:328-338"""
    k = int((recalls > recall).sum())
    return 0 if k == 0 else precisions[-k:].max()


def average_precisions(precisions, recalls, mode="voc2012"):
    """measurement.py:393-440 -> per-class AP list + their mean (last element)"""
    aps = []
    for p, r in zip(precisions, recalls):
        if mode in ("area", "smootharea"):
            q = np.maximum.accumulate(p[::-1])[::-1] if mode == "smootharea" else p
            ap = 0
            for k in range(len(q) - 1):
                ap += (r[k + 1] - r[k]) * ((q[k + 1] - q[k]) / 2 + q[k])
        else:
            grid = [0, 0.14, 0.29, 0.43, 0.57, 0.71, 1] if mode == "voc2012" else [k / 10 for k in range(11)]
            ap = 0
            for x in grid:
                ap += precision_at(p, r, x)
            ap = ap / len(grid)
        aps.append(ap)
    aps.append(sum(aps) / len(aps))
    return aps
