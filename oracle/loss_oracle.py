"""oracle/loss_oracle.py -- CPU restatement (plain torch, differentiable) of the targets + losses of the two-stage branch.

TEST INFRASTRUCTURE ONLY (tests/, smoke, bench cpu_baseline); nothing under sm3det_amd/ imports it.

Restated from the reference tree -- and PINNED to it: tests/test_oracle_losses.py runs the reference's own
``OrientedRPNHead.loss`` / ``RotatedBBoxHead.get_targets`` + ``loss`` (imported live through oracle/ref_heads.py) on the
same inputs and compares values and gradients:

* ``rpn_loss``  = RotatedRPNHead.loss + get_targets (mmrotate/models/dense_heads/rotated_rpn_head.py:152-372) over
  OrientedRPNHead._get_targets_single / loss_single (oriented_rpn_head.py:26-187), dense per-anchor targets and all;
* ``rcnn_loss`` = RotatedBBoxHead._get_target_single / get_targets / loss (roi_heads/bbox_heads/rotated_bbox_head.py:141-356).

Restated from mmdet 2.x (>= 2.25.1, requirements/runtime.txt:4) -- NOT vendored by the reference, **parity unpinned**:
``anchor_inside_flags`` / ``unmap`` / ``images_to_levels`` / ``multi_apply`` (mmdet/core), ``cross_entropy`` /
``binary_cross_entropy`` / ``smooth_l1_loss`` with ``weight_reduce_loss`` (mmdet/models/losses: ``sum(loss * weight) /
(avg_factor + eps)``, eps = float32 machine epsilon) and ``accuracy``.  ref_heads.py hands these same functions to the
reference classes as their mmdet imports, so the pinned part is everything AROUND the three loss formulas.
"""
from functools import partial

import torch
import torch.nn.functional as F

EPS = torch.finfo(torch.float32).eps


# ----------------------------------------------------------------------------------------------- mmdet stand-ins
def multi_apply(func, *args, **kwargs):
    pfunc = partial(func, **kwargs) if kwargs else func
    return tuple(map(list, zip(*map(pfunc, *args))))


def anchor_inside_flags(flat_anchors, valid_flags, img_shape, allowed_border=0):
    img_h, img_w = img_shape[:2]
    if allowed_border >= 0:
        return valid_flags & (flat_anchors[:, 0] >= -allowed_border) & (flat_anchors[:, 1] >= -allowed_border) & \
            (flat_anchors[:, 2] < img_w + allowed_border) & (flat_anchors[:, 3] < img_h + allowed_border)
    return valid_flags


def unmap(data, count, inds, fill=0):
    if data.dim() == 1:
        ret = data.new_full((count,), fill)
        ret[inds.type(torch.bool)] = data
    else:
        ret = data.new_full((count,) + data.size()[1:], fill)
        ret[inds.type(torch.bool), :] = data
    return ret


def images_to_levels(target, num_levels):
    target = torch.stack(target, 0)
    out, start = [], 0
    for n in num_levels:
        out.append(target[:, start:start + n])
        start += n
    return out


def weight_reduce_loss(loss, weight=None, reduction='mean', avg_factor=None):
    if weight is not None:
        loss = loss * weight
    if avg_factor is None:
        return loss.mean() if reduction == 'mean' else (loss.sum() if reduction == 'sum' else loss)
    if reduction == 'mean':
        return loss.sum() / (avg_factor + EPS)
    if reduction == 'none':
        return loss
    raise ValueError('avg_factor can not be used with reduction="sum"')


def _expand_onehot_labels(labels, label_weights, label_channels, ignore_index):
    bin_labels = labels.new_full((labels.size(0), label_channels), 0)
    valid_mask = (labels >= 0) & (labels != ignore_index)
    inds = torch.nonzero(valid_mask & (labels < label_channels), as_tuple=False)
    if inds.numel() > 0:
        bin_labels[inds, labels[inds]] = 1
    valid_mask = valid_mask.view(-1, 1).expand(labels.size(0), label_channels).float()
    if label_weights is None:
        w = valid_mask
    else:
        w = label_weights.view(-1, 1).repeat(1, label_channels) * valid_mask
    return bin_labels, w, valid_mask


class CrossEntropyLoss(torch.nn.Module):
    """mmdet CrossEntropyLoss for use_sigmoid in {False, True} (use_mask / class_weight / avg_non_ignore unused here)"""

    def __init__(self, use_sigmoid=False, use_mask=False, reduction='mean', class_weight=None, ignore_index=None,
                 loss_weight=1.0, avg_non_ignore=False):
        super().__init__()
        assert not use_mask and class_weight is None and not avg_non_ignore
        self.use_sigmoid, self.reduction, self.loss_weight = use_sigmoid, reduction, loss_weight
        self.ignore_index = -100 if ignore_index is None else ignore_index

    def forward(self, cls_score, label, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        reduction = reduction_override if reduction_override else self.reduction
        if self.use_sigmoid:
            if cls_score.dim() != label.dim():
                label, weight, _ = _expand_onehot_labels(label, weight, cls_score.size(-1), self.ignore_index)
            loss = F.binary_cross_entropy_with_logits(cls_score, label.float(), reduction='none')
            return self.loss_weight * weight_reduce_loss(loss, weight.float(), reduction, avg_factor)
        loss = F.cross_entropy(cls_score, label, reduction='none', ignore_index=self.ignore_index)
        if weight is not None:
            weight = weight.float()
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction, avg_factor)


class SmoothL1Loss(torch.nn.Module):
    def __init__(self, beta=1.0, reduction='mean', loss_weight=1.0):
        super().__init__()
        self.beta, self.reduction, self.loss_weight = beta, reduction, loss_weight

    def forward(self, pred, target, weight=None, avg_factor=None, reduction_override=None, **kwargs):
        reduction = reduction_override if reduction_override else self.reduction
        if target.numel() == 0:
            return pred.sum() * 0
        diff = torch.abs(pred - target)
        loss = torch.where(diff < self.beta, 0.5 * diff * diff / self.beta, diff - 0.5 * self.beta)
        return self.loss_weight * weight_reduce_loss(loss, weight, reduction, avg_factor)


def accuracy(pred, target, topk=1, thresh=None):
    assert topk == 1 and thresh is None
    if pred.size(0) == 0:
        return pred.new_tensor(0.0)
    _, lab = pred.topk(1, dim=1)
    return lab.t().eq(target.view(1, -1)).reshape(-1).float().sum(0, keepdim=True).mul_(100.0 / pred.size(0))[0]


def build_loss(cfg):
    cfg = dict(cfg)
    return {'CrossEntropyLoss': CrossEntropyLoss, 'SmoothL1Loss': SmoothL1Loss}[cfg.pop('type')](**cfg)


# ----------------------------------------------------------------------------------------------- the restatements
def rpn_loss(cls_scores, bbox_preds, mlvl_anchors, inside_flags, gt_bboxes, pos_inds, neg_inds, means, stds,
             beta=1.0 / 9.0, loss_weight_cls=1.0, loss_weight_bbox=1.0, pos_weight=-1.0, assigned_gt=None,
             assign_cfg=None):
    """cls_scores[l] (B, A, H, W), bbox_preds[l] (B, 6A, H, W); mlvl_anchors[l] (H W A, 4); inside_flags (total,) bool;
    gt_bboxes list of (k_i, 5); pos_inds / neg_inds lists of FLAT anchor indices per image (the sampler's choice);
    assigned_gt list of (len(pos_inds[i]),) gt indices, or None to recompute the assignment with oracle/assign_oracle.py.
    Returns (loss_cls per level, loss_bbox per level) -- the dense pipeline of the reference, restated."""
    from oracle import assign_oracle, rpn_oracle
    B = cls_scores[0].shape[0]
    num_level_anchors = [a.shape[0] for a in mlvl_anchors]
    flat = torch.cat(mlvl_anchors)
    total = flat.shape[0]
    all_labels, all_lw, all_bt, all_bw, npos, nneg = [], [], [], [], 0, 0
    for i in range(B):
        labels = torch.full((total,), 1, dtype=torch.long)       # num_classes = 1 -> background label 1
        lw = torch.zeros(total)
        bt, bw = torch.zeros(total, 6), torch.zeros(total, 6)
        p, n = pos_inds[i].long(), neg_inds[i].long()
        assert bool(inside_flags[p].all()) and bool(inside_flags[n].all())
        if p.numel():
            if assigned_gt is not None:
                g = assigned_gt[i].long()
            else:
                hb = rpn_oracle.obb2xyxy_le90(gt_bboxes[i])
                gi, _, _, _ = assign_oracle.max_iou_assign(flat[inside_flags].numpy(), hb.numpy(), False, **assign_cfg)
                full = torch.full((total,), -1, dtype=torch.long)
                full[inside_flags] = torch.from_numpy(gi)
                g = full[p] - 1
                assert bool((g >= 0).all())
            bt[p] = rpn_oracle.midpoint_bbox2delta(flat[p], gt_bboxes[i][g], means, stds)
            bw[p] = 1.0
            labels[p] = 0
            lw[p] = 1.0 if pos_weight <= 0 else pos_weight
        if n.numel():
            lw[n] = 1.0
        all_labels.append(labels); all_lw.append(lw); all_bt.append(bt); all_bw.append(bw)
        npos += max(int(p.numel()), 1)
        nneg += max(int(n.numel()), 1)
    avg = npos + nneg
    lab_l = images_to_levels(all_labels, num_level_anchors)
    lw_l = images_to_levels(all_lw, num_level_anchors)
    bt_l = images_to_levels(all_bt, num_level_anchors)
    bw_l = images_to_levels(all_bw, num_level_anchors)
    ce = CrossEntropyLoss(use_sigmoid=True, loss_weight=loss_weight_cls)
    sl = SmoothL1Loss(beta=beta, loss_weight=loss_weight_bbox)
    out_c, out_b = [], []
    for l in range(len(cls_scores)):
        cs = cls_scores[l].permute(0, 2, 3, 1).reshape(-1, 1)
        bp = bbox_preds[l].permute(0, 2, 3, 1).reshape(-1, 6)
        out_c.append(ce(cs, lab_l[l].reshape(-1), lw_l[l].reshape(-1), avg_factor=avg))
        out_b.append(sl(bp, bt_l[l].reshape(-1, 6), bw_l[l].reshape(-1, 6), avg_factor=avg))
    return out_c, out_b


def rcnn_loss(cls_score, bbox_pred, pos_bboxes, neg_bboxes, pos_gt_bboxes, pos_gt_labels, num_classes, means, stds,
              edge_swap=True, proj_xy=True, beta=1.0, loss_weight_cls=1.0, loss_weight_bbox=1.0, pos_weight=-1.0):
    """lists per image of pos_bboxes (p_i,5) / neg_bboxes (n_i,5) / pos_gt_bboxes (p_i,5) / pos_gt_labels (p_i,);
    cls_score (sum(p_i+n_i), C+1), bbox_pred (.., 5) ordered [image 0 positives, image 0 negatives, image 1 ...].
    Returns dict(loss_cls, acc, loss_bbox)."""
    from oracle import rpn_oracle
    labels, lw, bt, bw = [], [], [], []
    for pb, nb, pg, pl in zip(pos_bboxes, neg_bboxes, pos_gt_bboxes, pos_gt_labels):
        npos, nneg = pb.shape[0], nb.shape[0]
        n = npos + nneg
        lab = torch.full((n,), num_classes, dtype=torch.long)
        w = torch.zeros(n)
        t, tw = torch.zeros(n, 5), torch.zeros(n, 5)
        if npos:
            lab[:npos] = pl
            w[:npos] = 1.0 if pos_weight <= 0 else pos_weight
            t[:npos] = rpn_oracle.xywha_bbox2delta(pb, pg, means, stds, None, edge_swap, proj_xy)
            tw[:npos] = 1
        if nneg:
            w[-nneg:] = 1.0
        labels.append(lab); lw.append(w); bt.append(t); bw.append(tw)
    labels, lw, bt, bw = torch.cat(labels), torch.cat(lw), torch.cat(bt), torch.cat(bw)
    out = {}
    avg = max(float((lw > 0).sum()), 1.0)
    out['loss_cls'] = CrossEntropyLoss(loss_weight=loss_weight_cls)(cls_score, labels, lw, avg_factor=avg)
    out['acc'] = accuracy(cls_score, labels)
    pos = (labels >= 0) & (labels < num_classes)
    if pos.any():
        out['loss_bbox'] = SmoothL1Loss(beta=beta, loss_weight=loss_weight_bbox)(
            bbox_pred.view(bbox_pred.size(0), 5)[pos], bt[pos], bw[pos], avg_factor=bt.size(0))
    else:
        out['loss_bbox'] = bbox_pred[pos].sum()
    return out
