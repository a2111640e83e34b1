"""Loss closures of the four YOLO versions restated on torch CPU (float64 by default,
autograd-able), following the reference line by line:

  v3   yolov3/losses/loss.py:9-37 (cal_iou), :40-164 (wrap_yolo_loss)
  v2   yolov2/losses/loss.py:40-137
  v4   yolov4/losses/loss.py:10-61 (cal_iou + CIoU), :64-169
  v1.5 yolov1_5/losses/loss.py:40-118

TensorFlow autodiff conventions that matter are made explicit:
tf.maximum / tf.minimum send the gradient of a tie to the FIRST operand (`_tf_max/_tf_min`
select with >= / <=), clip_by_value passes gradient on the closed interval (torch.clamp does
too), argmax / one_hot / comparisons are constants.
"""
import math

import torch

EPSILON = 1e-07


def _tf_max(x, y):
    y = torch.as_tensor(y, dtype=x.dtype) if not torch.is_tensor(y) else y
    return torch.where(x >= y, x, y.expand_as(x) if y.dim() == 0 else y)


def _tf_min(x, y):
    y = torch.as_tensor(y, dtype=x.dtype) if not torch.is_tensor(y) else y
    return torch.where(x <= y, x, y.expand_as(x) if y.dim() == 0 else y)


def _bc(a, b):
    return torch.broadcast_tensors(a, b)


def cal_iou(xywh_true, xywh_pred, grid_shape, return_ciou=False):
    """yolov3/losses/loss.py:9-37; with return_ciou: yolov4/losses/loss.py:10-61."""
    gs = torch.tensor([grid_shape[1], grid_shape[0]], dtype=xywh_true.dtype)  # grid_shape[::-1]
    xy_true = xywh_true[..., 0:2] / gs
    wh_true = xywh_true[..., 2:4]
    xy_pred = xywh_pred[..., 0:2] / gs
    wh_pred = xywh_pred[..., 2:4]

    half_wh_true = wh_true / 2.
    mins_true = xy_true - half_wh_true
    maxes_true = xy_true + half_wh_true
    half_wh_pred = wh_pred / 2.
    mins_pred = xy_pred - half_wh_pred
    maxes_pred = xy_pred + half_wh_pred

    intersect_mins = _tf_max(*_bc(mins_pred, mins_true))
    intersect_maxes = _tf_min(*_bc(maxes_pred, maxes_true))
    intersect_wh = _tf_max(intersect_maxes - intersect_mins, 0.)
    intersect_areas = intersect_wh[..., 0] * intersect_wh[..., 1]

    true_areas = wh_true[..., 0] * wh_true[..., 1]
    pred_areas = wh_pred[..., 0] * wh_pred[..., 1]
    union_areas = pred_areas + true_areas - intersect_areas
    iou_scores = intersect_areas / (union_areas + EPSILON)
    if not return_ciou:
        return iou_scores

    enclose_mins = _tf_min(*_bc(mins_pred, mins_true))
    enclose_maxes = _tf_max(*_bc(maxes_pred, maxes_true))
    enclose_wh = enclose_maxes - enclose_mins
    enclose_c2 = enclose_wh[..., 0] ** 2 + enclose_wh[..., 1] ** 2
    p_rho2 = (xy_true[..., 0] - xy_pred[..., 0]) ** 2 + (xy_true[..., 1] - xy_pred[..., 1]) ** 2
    atan_true = torch.atan(wh_true[..., 0] / (wh_true[..., 1] + EPSILON))
    atan_pred = torch.atan(wh_pred[..., 0] / (wh_pred[..., 1] + EPSILON))
    v_nu = 4.0 / (math.pi ** 2) * (atan_true - atan_pred) ** 2
    a_alpha = v_nu / (1 - iou_scores + v_nu)
    ciou_scores = iou_scores - p_rho2 / enclose_c2 - a_alpha * v_nu
    return iou_scores, ciou_scores


def _one_hot_argmax(iou_scores, depth):
    idx = torch.argmax(iou_scores.detach(), dim=-1)  # first maximal index, like tf.argmax
    return torch.nn.functional.one_hot(idx, depth).to(iou_scores.dtype)


def _decisions(iou_own, dec, bbox_num, stats, ignore_thresh=None, truth_thresh=None):
    """The discrete decisions of a loss: the responsible anchor (argmax IoU) and the ignore / truth masks (IoU against a
    threshold). dec = None: taken from this execution's own IoUs. dec = int tensor [..., 2] as yolo_loss_fwd_bwd exports it
    ([0] responsible anchor, [1] bit b = IoU_b < ignore_thresh, bit 16 + b = IoU_b > truth_thresh): taken from ANOTHER
    execution (the device) -- the loss-side twin of layers.leaky_masked: two executions whose IoUs differ by rounding (or
    whose fp32 / fp64 evaluation of one tiny intersection lands on different sides of zero) pick different winners among
    tied anchors, and one different winner moves a head's gradient tensor by O(1). Returns (one-hot of the responsible
    anchor, a surrogate IoU tensor whose comparisons with the two thresholds reproduce the masks). stats["disagree"]
    records how far from a tie / from the threshold the own IoUs were where the decisions differ (the caller asserts that
    this is rounding-sized)."""
    if dec is None:
        idx = torch.argmax(iou_own, dim=-1)
        return torch.nn.functional.one_hot(idx, bbox_num).to(iou_own.dtype), iou_own
    dec = dec.reshape(*iou_own.shape[:-1], 2).long()
    idx = dec[..., 0]
    bits = dec[..., 1]
    ar = torch.arange(bbox_num)
    ign = ((bits.unsqueeze(-1) >> ar) & 1).bool()
    tru = ((bits.unsqueeze(-1) >> (16 + ar)) & 1).bool()
    lo = -1.0
    mid = ignore_thresh if ignore_thresh is not None else 0.0
    src = torch.full_like(iou_own, float(mid))            # mid = ignore_thresh (<= truth_thresh): in neither mask
    if truth_thresh is not None and truth_thresh < 1:
        src = torch.where(tru, torch.full_like(iou_own, float(truth_thresh) + 1.0), src)
    if ignore_thresh is not None:
        src = torch.where(ign, torch.full_like(iou_own, lo), src)
    if stats is not None:
        own = torch.argmax(iou_own, dim=-1)
        gap = (iou_own.gather(-1, own.unsqueeze(-1)) - iou_own.gather(-1, idx.unsqueeze(-1))).squeeze(-1)
        d = float(gap[own != idx].max()) if bool((own != idx).any()) else 0.0
        # how MANY decisions were forced against this execution's own (the caller prints and bounds them)
        stats["cells"] = stats.get("cells", 0) + own.numel()
        stats["n_anchor"] = stats.get("n_anchor", 0) + int((own != idx).sum())
        if ignore_thresh is not None:
            bad = (iou_own < ignore_thresh) != ign
            stats["n_ignore"] = stats.get("n_ignore", 0) + int(bad.sum())
            if bool(bad.any()):
                d = max(d, float((iou_own[bad] - ignore_thresh).abs().max()))
        if truth_thresh is not None and truth_thresh < 1:
            bad = (iou_own > truth_thresh) != tru
            stats["n_truth"] = stats.get("n_truth", 0) + int(bad.sum())
            if bool(bad.any()):
                d = max(d, float((iou_own[bad] - truth_thresh).abs().max()))
        stats["disagree"] = max(stats.get("disagree", 0.0), d)
    return torch.nn.functional.one_hot(idx, bbox_num).to(iou_own.dtype), src


def _sum_mean0(t):
    """tf.reduce_sum(tf.reduce_mean(t, axis=0))"""
    return t.mean(dim=0).sum()


def wrap_yolo_loss_v3(grid_shape, bbox_num, class_num, anchors=None, binary_weight=1,
                      loss_weight=(1, 1, 1, 1), ignore_thresh=.6, use_focal_loss=False,
                      focal_loss_gamma=2, use_scale=True, parts=False):
    def yolo_loss(y_true, y_pred, decide_with=None, stats=None):
        dt = y_pred.dtype
        panchors = 1 if anchors is None else torch.tensor(anchors, dtype=dt).reshape(1, 1, 1, bbox_num, 2)
        y_true_ = y_true.reshape(-1, *grid_shape, 1, 5 + class_num).to(dt)
        y_pred_ = y_pred.reshape(-1, *grid_shape, bbox_num, 5 + class_num)
        xywh_true = y_true_[..., :4]
        xywh_pred = y_pred_[..., :4]
        iou_scores = cal_iou(xywh_true, xywh_pred, grid_shape).detach()
        response_mask, iou_scores = _decisions(iou_scores, decide_with, bbox_num, stats, ignore_thresh)
        has_obj_mask = y_true_[..., 4] * response_mask
        has_obj_mask_exp = has_obj_mask.unsqueeze(-1)
        no_obj_mask = (iou_scores < ignore_thresh).to(dt)
        no_obj_mask = (1 - has_obj_mask) * no_obj_mask

        xy_true = y_true_[..., 0:2]
        xy_pred = y_pred_[..., 0:2]
        wh_true = _tf_max(y_true_[..., 2:4] / panchors, EPSILON)
        wh_pred = y_pred_[..., 2:4] / panchors
        wh_true = torch.log(wh_true)
        wh_pred = torch.log(wh_pred)
        c_pred = y_pred_[..., 4]
        box_loss_scale = (2 - y_true_[..., 2:3] * y_true_[..., 3:4]) if use_scale else 1

        xy_loss = _sum_mean0(has_obj_mask_exp * box_loss_scale * (xy_true - xy_pred) ** 2)
        wh_loss = _sum_mean0(has_obj_mask_exp * box_loss_scale * (wh_true - wh_pred) ** 2)
        if use_focal_loss:
            c_pred = torch.clamp(c_pred, EPSILON, 1 - EPSILON)
            has_obj_c_loss = -_sum_mean0(has_obj_mask * ((1 - c_pred) ** focal_loss_gamma) * torch.log(c_pred))
            no_obj_c_loss = -_sum_mean0(no_obj_mask * (c_pred ** focal_loss_gamma) * torch.log(1 - c_pred))
        else:
            has_obj_c_loss = _sum_mean0(has_obj_mask * (1 - c_pred) ** 2)
            no_obj_c_loss = _sum_mean0(no_obj_mask * (0 - c_pred) ** 2)
        c_loss = has_obj_c_loss + binary_weight * no_obj_c_loss

        p_true = y_true_[..., -class_num:]
        p_pred = torch.clamp(y_pred_[..., -class_num:], EPSILON, 1 - EPSILON)
        p_loss = -_sum_mean0(has_obj_mask_exp * (p_true * torch.log(p_pred) + (1 - p_true) * torch.log(1 - p_pred)))
        regularizer = _sum_mean0(wh_pred ** 2) * 0.01
        loss = (loss_weight[0] * xy_loss + loss_weight[1] * wh_loss + loss_weight[2] * c_loss
                + loss_weight[3] * p_loss + regularizer)
        if parts:
            return loss, dict(xy=xy_loss, wh=wh_loss, conf_obj=has_obj_c_loss, conf_noobj=no_obj_c_loss,
                              prob=p_loss, reg=regularizer / 0.01)
        return loss
    return yolo_loss


def wrap_yolo_loss_v2(grid_shape, bbox_num, class_num, anchors, binary_weight=1, loss_weight=(1, 1, 1, 1),
                      ignore_thresh=.6):
    def yolo_loss(y_true, y_pred, decide_with=None, stats=None):
        dt = y_pred.dtype
        panchors = torch.tensor(anchors, dtype=dt).reshape(1, 1, 1, bbox_num, 2)
        y_true_ = y_true.reshape(-1, *grid_shape, 1, 5 + class_num).to(dt)
        y_pred_ = y_pred.reshape(-1, *grid_shape, bbox_num, 5 + class_num)
        iou_scores = cal_iou(y_true_[..., :4], y_pred_[..., :4], grid_shape).detach()
        response_mask, iou_scores = _decisions(iou_scores, decide_with, bbox_num, stats, ignore_thresh)
        has_obj_mask = y_true_[..., 4] * response_mask
        has_obj_mask_exp = has_obj_mask.unsqueeze(-1)
        no_obj_mask = (1 - has_obj_mask) * (iou_scores < ignore_thresh).to(dt)
        xy_true, xy_pred = y_true_[..., 0:2], y_pred_[..., 0:2]
        wh_true = torch.log(_tf_max(y_true_[..., 2:4] / panchors, EPSILON))
        wh_pred = torch.log(y_pred_[..., 2:4] / panchors)
        c_pred = y_pred_[..., 4]
        box_loss_scale = 2 - y_true_[..., 2:3] * y_true_[..., 3:4]
        xy_loss = _sum_mean0(has_obj_mask_exp * box_loss_scale * (xy_true - xy_pred) ** 2)
        wh_loss = _sum_mean0(has_obj_mask_exp * box_loss_scale * (wh_true - wh_pred) ** 2)
        has_obj_c_loss = _sum_mean0(has_obj_mask * (1 - c_pred) ** 2)
        no_obj_c_loss = _sum_mean0(no_obj_mask * (0 - c_pred) ** 2)
        c_loss = has_obj_c_loss + binary_weight * no_obj_c_loss
        p_true = y_true_[..., -class_num:]
        p_pred = torch.clamp(y_pred_[..., -class_num:], EPSILON, 1 - EPSILON)
        p_loss = -_sum_mean0(has_obj_mask_exp * (p_true * torch.log(p_pred)))
        regularizer = _sum_mean0(wh_pred ** 2) * 0.01
        return (loss_weight[0] * xy_loss + loss_weight[1] * wh_loss + loss_weight[2] * c_loss
                + loss_weight[3] * p_loss + regularizer)
    return yolo_loss


def wrap_yolo_loss_v4(grid_shape, bbox_num, class_num, anchors=None, binary_weight=1, loss_weight=(1, 1, 1),
                      wh_reg_weight=0.01, ignore_thresh=.6, truth_thresh=1, label_smooth=0, focal_loss_gamma=2):
    def yolo_loss(y_true, y_pred, decide_with=None, stats=None):
        dt = y_pred.dtype
        panchors = 1 if anchors is None else torch.tensor(anchors, dtype=dt).reshape(1, 1, 1, bbox_num, 2)
        y_true_ = y_true.reshape(-1, *grid_shape, 1, 5 + class_num).to(dt)
        y_pred_ = y_pred.reshape(-1, *grid_shape, bbox_num, 5 + class_num)
        iou_scores, ciou_scores = cal_iou(y_true_[..., :4], y_pred_[..., :4], grid_shape, return_ciou=True)
        iou_const = iou_scores.detach()
        response_mask, iou_const = _decisions(iou_const, decide_with, bbox_num, stats, ignore_thresh, truth_thresh)
        has_obj_mask = y_true_[..., 4] * response_mask
        if truth_thresh < 1:
            truth_mask = (iou_const > truth_thresh).to(dt)
            has_obj_mask = has_obj_mask + truth_mask * (1 - has_obj_mask)
        has_obj_mask_exp = has_obj_mask.unsqueeze(-1)
        no_obj_mask = (1 - has_obj_mask) * (iou_const < ignore_thresh).to(dt)
        box_loss = _sum_mean0(has_obj_mask * (1 - ciou_scores))
        c_pred = torch.clamp(y_pred_[..., 4], EPSILON, 1 - EPSILON)
        if label_smooth > 0:
            obj_error = torch.abs(1 - label_smooth - c_pred)
            no_obj_error = torch.abs(label_smooth - c_pred)
        else:
            obj_error = 1 - c_pred
            no_obj_error = c_pred
        has_obj_c_loss = -_sum_mean0(has_obj_mask * (obj_error ** focal_loss_gamma) * torch.log(1 - obj_error))
        no_obj_c_loss = -_sum_mean0(no_obj_mask * (no_obj_error ** focal_loss_gamma) * torch.log(1 - no_obj_error))
        c_loss = has_obj_c_loss + binary_weight * no_obj_c_loss
        p_true = y_true_[..., -class_num:]
        p_pred = torch.clamp(y_pred_[..., -class_num:], EPSILON, 1 - EPSILON)
        p_loss = -_sum_mean0(has_obj_mask_exp * (p_true * torch.log(p_pred) + (1 - p_true) * torch.log(1 - p_pred)))
        wh_pred = torch.log(y_pred_[..., 2:4] / panchors)
        wh_reg = _sum_mean0(wh_pred ** 2)
        return (loss_weight[0] * box_loss + loss_weight[1] * c_loss + loss_weight[2] * p_loss
                + wh_reg_weight * wh_reg)
    return yolo_loss


def wrap_yolo_loss_v1(grid_shape, bbox_num, class_num, binary_weight=1, loss_weight=(1, 1, 1, 1)):
    def yolo_loss(y_true, y_pred, decide_with=None, stats=None):
        dt = y_pred.dtype
        y_true = y_true.to(dt)
        xywhc_true = y_true[..., :-class_num].reshape(-1, *grid_shape, 1, 5)
        xywhc_pred = y_pred[..., :-class_num].reshape(-1, *grid_shape, bbox_num, 5)
        iou_scores = cal_iou(xywhc_true, xywhc_pred, grid_shape)   # differentiated (loss.py:86-91)
        response_mask, _ = _decisions(iou_scores.detach(), decide_with, bbox_num, stats)
        response_mask_exp = response_mask.unsqueeze(-1)
        has_obj_mask = xywhc_true[..., 4]
        has_obj_mask_exp = has_obj_mask.unsqueeze(-1)
        no_obj_mask = 1 - has_obj_mask * response_mask
        xy_true, xy_pred = xywhc_true[..., 0:2], xywhc_pred[..., 0:2]
        wh_true = _tf_max(xywhc_true[..., 2:4], EPSILON)
        wh_pred = _tf_max(xywhc_pred[..., 2:4], EPSILON)
        c_pred = xywhc_pred[..., 4]
        xy_loss = _sum_mean0(has_obj_mask_exp * response_mask_exp * (xy_true - xy_pred) ** 2)
        wh_loss = _sum_mean0(has_obj_mask_exp * response_mask_exp * (torch.sqrt(wh_true) - torch.sqrt(wh_pred)) ** 2)
        has_obj_c_loss = _sum_mean0(has_obj_mask * response_mask * (iou_scores - c_pred) ** 2)
        no_obj_c_loss = _sum_mean0(no_obj_mask * (0 - c_pred) ** 2)
        c_loss = has_obj_c_loss + binary_weight * no_obj_c_loss
        p_true = y_true[..., -class_num:].reshape(-1, *grid_shape, class_num)
        p_pred = torch.clamp(y_pred[..., -class_num:].reshape(-1, *grid_shape, class_num), EPSILON, 1 - EPSILON)
        p_loss = -_sum_mean0(has_obj_mask * p_true * torch.log(p_pred))
        return (loss_weight[0] * xy_loss + loss_weight[1] * wh_loss + loss_weight[2] * c_loss
                + loss_weight[3] * p_loss)
    return yolo_loss
