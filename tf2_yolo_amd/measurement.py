"""Drop-in for the reference's utils/measurement.py (create_score_mat :16-150, PRfunc :153-447), on the GPU.

Same names, arguments, defaults and return types (pandas DataFrames; PRfunc.precisions / .recalls lists of
float64 arrays). Per image the pipeline is decode -> NMS -> IoU matching, all in HIP kernels (decode_nms.hip,
measure.hip); the precision / recall curve of a class is one more kernel over all its detections. The host
keeps what the reference keeps in Python lists: which rows belong to which image and class.

Differences, all in corners the reference leaves undefined or broken:
  * exact ties of the joint confidence: the later detection sorts first (np.argsort is unstable);
  * a class without any detection: the reference re-uses `num_tp` of the PREVIOUS class (or raises
    NameError for class 0); here its curve is the single point (0, 0) -- (0, nan) if it also has no
    ground truth;
  * a class with detections but no ground truth: the reference divides by zero (ZeroDivisionError); here
    recall is nan and precision follows the formulas with num_tp = 0.
"""
import warnings

import numpy as np
import torch

from . import ops, tools
from ._lib import YoloHipError


def _per_image_rows(y_true, y_pred, class_num, conf_threshold, nms_mode, nms_threshold, nms_sigma, version):
    """decode (+ NMS) of one image -> float64 CUDA tensors (n,7): ground truth rows, detection rows"""
    gt = tools.decode_device(y_true, class_num=class_num, version=version)
    det = tools.decode_device(*y_pred, class_num=class_num, threshold=conf_threshold, version=version)
    if nms_mode > 0 and det.shape[0] > 0:
        if nms_mode == 1:
            det = tools.nms(det, class_num, nms_threshold)
        elif nms_mode == 2:
            det = tools.soft_nms(det, class_num, nms_threshold, conf_threshold, nms_sigma)
        elif nms_mode == 3:
            det = tools.nms(det, class_num, nms_threshold, 2)
    return gt, det


def _images(y_trues, y_preds):
    for i_label in range(len(y_trues)):
        yield y_trues[i_label], [y_preds[j][i_label] for j in range(len(y_preds))]


def create_score_mat(y_trues, *y_preds, class_names=[], conf_threshold=0.5, nms_mode=0, nms_threshold=0.5,
                     nms_sigma=0.5, iou_threshold=0.5, precision_mode=2, version=3):
    """Score table (pandas.DataFrame: precision, recall, F1-score, gts, dets per class)."""
    import pandas as pd
    if not torch.cuda.is_available():
        raise YoloHipError("tf2_yolo_amd.measurement needs a HIP device: there is no CPU fallback")
    class_num = len(class_names)
    counts = torch.zeros((max(class_num, 1), 4), device="cuda", dtype=torch.int64)
    for y_true, y_pred in _images(y_trues, y_preds):
        gt, det = _per_image_rows(y_true, y_pred, class_num, conf_threshold, nms_mode, nms_threshold, nms_sigma, version)
        if class_num:
            ops.match_detections(gt, det, class_num, iou_threshold, counts)
    c = counts.cpu().numpy()[:class_num].astype(np.float64)   # dets, gts, tpp, tp
    dets, gts, tpp, tp = c[:, 0], c[:, 1], c[:, 2], c[:, 3]
    denom_p = dets - (tpp - tp) if precision_mode == 1 else dets
    num_p = tpp if precision_mode == 0 else tp
    with np.errstate(divide="ignore", invalid="ignore"):
        precision = np.true_divide(num_p, denom_p)
        recall = np.true_divide(tp, gts)
        f1 = (2 * precision * recall) / (precision + recall)
    table = pd.DataFrame({"precision": precision, "recall": recall, "F1-score": f1,
                          "gts": gts.astype("int"), "dets": dets.astype("int")})
    table.index = class_names
    return table


class PRfunc(object):
    """Precision-recall function: call with a recall value (and class index) to get a precision value."""

    def __init__(self, y_trues, *y_preds, class_names=[], conf_threshold=0.05, nms_mode=1, nms_threshold=0.5,
                 nms_sigma=0.5, iou_threshold=0.5, precision_mode=2, max_per_img=100, version=3):
        if not torch.cuda.is_available():
            raise YoloHipError("tf2_yolo_amd.measurement needs a HIP device: there is no CPU fallback")
        class_num = len(class_names)
        self.class_num = class_num
        self.class_names = class_names
        gts = torch.zeros(max(class_num, 1), device="cuda", dtype=torch.int64)   # running ground-truth count
        scratch = torch.zeros((max(class_num, 1), 4), device="cuda", dtype=torch.int64)
        joint_l, gid_l, matched_l, cls_l = [], [], [], []
        for y_true, y_pred in _images(y_trues, y_preds):
            gt, det = _per_image_rows(y_true, y_pred, class_num, conf_threshold, nms_mode, nms_threshold,
                                      nms_sigma, version)
            if not class_num:
                continue
            scratch.zero_()
            best_gt, matched, _ = ops.match_detections(gt, det, class_num, iou_threshold, scratch)
            if det.shape[0]:
                cls = det[:, 5].to(torch.int64)
                ok = (cls >= 0) & (cls < class_num)
                clsc = cls.clamp(0, class_num - 1)
                joint = (det[:, 4] * det[:, 6]).contiguous()
                # ids are global per class: index inside the image's class subset + ground truths seen so far
                gid = torch.where(scratch[clsc, 1] > 0, best_gt.to(torch.int64) + gts[clsc], torch.zeros_like(cls))
                keep = ok
                if max_per_img is not None:
                    rank = ops.rank_desc(joint, cls.to(torch.int32))
                    keep = keep & (rank < max_per_img)
                joint_l.append(joint[keep]); gid_l.append(gid[keep].to(torch.int32))
                matched_l.append(matched[keep]); cls_l.append(cls[keep])
            gts += scratch[:, 1]
        gts_h = gts.cpu().numpy()
        if joint_l:
            joint_a, gid_a = torch.cat(joint_l), torch.cat(gid_l)
            matched_a, cls_a = torch.cat(matched_l), torch.cat(cls_l)
        self.precisions, self.recalls = [], []
        for c in range(class_num):
            sel = (cls_a == c) if joint_l else None
            n = int(sel.sum().item()) if sel is not None else 0
            num_gts = int(gts_h[c])
            if n == 0:
                self.precisions.append(np.array([0.0]))
                self.recalls.append(np.array([0.0 if num_gts > 0 else np.nan]))
                continue
            if num_gts == 0:   # every detection is a false positive; recall undefined
                self.precisions.append(np.zeros(n + 1))
                self.recalls.append(np.full(n + 1, np.nan))
                continue
            p, r = ops.pr_curve(joint_a[sel].contiguous(), gid_a[sel].contiguous(), matched_a[sel].contiguous(),
                                num_gts, precision_mode)
            self.precisions.append(p.cpu().numpy())
            self.recalls.append(r.cpu().numpy())

    def __call__(self, recall, class_idx=0):
        if class_idx >= self.class_num:
            raise IndexError("Class index out of range")
        precisions = self.precisions[class_idx]
        recalls = self.recalls[class_idx]
        pc_idx = (recalls > recall).sum()
        if pc_idx == 0:
            return 0
        return precisions[-pc_idx:].max()

    def plot_pr_curve(self, class_idx=-1, smooth=False, figsize=None, return_fig=False):
        """Plot PR curve(s) with matplotlib (class_idx = -1: all classes; smooth: interpolated precision)."""
        import matplotlib.pyplot as plt
        if class_idx >= self.class_num:
            raise IndexError("Class index out of range")
        sl = slice(class_idx, class_idx + 1) if class_idx >= 0 else slice(None)
        fig = plt.figure(figsize=figsize)
        for precision, recall in zip(self.precisions[sl], self.recalls[sl]):
            if smooth:
                precision = np.maximum.accumulate(precision[::-1])[::-1]
            plt.plot(recall, precision)
        plt.legend(self.class_names[sl])
        plt.title("PR curve")
        plt.xlabel("recall")
        plt.ylabel("precision")
        plt.xlim(-0.05, 1.05)
        plt.ylim(-0.05, 1.05)
        if return_fig:
            return fig
        plt.show()

    def get_map(self, mode="voc2012"):
        """mAP table (pandas.DataFrame); mode in "voc2007", "voc2012", "area", "smootharea"."""
        import pandas as pd
        aps = [0 for _ in range(self.class_num)]
        if mode in ("area", "smootharea"):
            for c in range(self.class_num):
                precisions = self.precisions[c]
                if mode == "smootharea":
                    precisions = np.maximum.accumulate(precisions[::-1])[::-1]
                recalls = self.recalls[c]
                for i in range(len(precisions) - 1):   # trapezoids, summed left to right like the reference
                    aps[c] += (recalls[i + 1] - recalls[i]) * ((precisions[i + 1] - precisions[i]) / 2 + precisions[i])
        else:
            if mode == "voc2012":
                recall_list = [0, 0.14, 0.29, 0.43, 0.57, 0.71, 1]
            elif mode == "voc2007":
                recall_list = [i / 10 for i in range(0, 11)]
            else:
                raise ValueError(f"Invalid mode: {mode}")
            for c in range(self.class_num):
                for rc in recall_list:
                    aps[c] += self(rc, c)
            aps = [ap / len(recall_list) for ap in aps]
        aps.append(sum(aps) / len(aps))
        table = pd.DataFrame(aps)
        table.columns = ["ap"]
        table.index = list(self.class_names) + ["mAP"]
        return table


class PR_func(PRfunc):
    def __init__(self, *args, **kwargs):
        warnings.warn("`PR_func` is deprecated and renamed to `PRfunc`.", Warning)
        super().__init__(*args, **kwargs)
