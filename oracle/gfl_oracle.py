"""oracle/gfl_oracle.py -- plain-torch restatement of the GFL head's conv towers (mmdet 2.x
``mmdet/models/dense_heads/gfl_head.py`` ``_init_layers`` / ``forward_single``: 4 x [Conv2d 3x3 no bias, GroupNorm(32),
ReLU] per tower, ``gfl_cls`` / ``gfl_reg`` 3x3 convs, per-level ``Scale``) over an mmdet-schema state_dict.

TEST INFRASTRUCTURE ONLY.  **PARITY UNPINNED**: mmdet is not vendored under /root/reference (requirements/runtime.txt:4
`mmdet>=2.25.1,<3.0.0`; call site: the type string at local_configs/main_SM3Det.py:30), so there is neither source nor a
reference test to check this restatement against; it follows the published mmdet 2.25 implementation from memory."""
import torch.nn.functional as F


def tower(x, p, prefix, n):
    for i in range(n):
        x = F.conv2d(x, p[f'{prefix}.{i}.conv.weight'], None, padding=1)
        x = F.relu(F.group_norm(x, 32, p[f'{prefix}.{i}.gn.weight'], p[f'{prefix}.{i}.gn.bias'], 1e-5))
    return x


def forward_single(x, p, level, stacked_convs=4):
    cls_feat = tower(x, p, 'cls_convs', stacked_convs)
    reg_feat = tower(x, p, 'reg_convs', stacked_convs)
    cls_score = F.conv2d(cls_feat, p['gfl_cls.weight'], p['gfl_cls.bias'], padding=1)
    bbox_pred = F.conv2d(reg_feat, p['gfl_reg.weight'], p['gfl_reg.bias'], padding=1) * p[f'scales.{level}.scale']
    return cls_score, bbox_pred.float()


def forward(feats, p, stacked_convs=4):
    outs = [forward_single(f, p, l, stacked_convs) for l, f in enumerate(feats)]
    return [o[0] for o in outs], [o[1] for o in outs]


# ---------------------------------------------------------------------------------------------------------------
# Loss side (round 4): ATSSAssigner + PseudoSampler + GFLHead.loss in mmdet's OWN control flow -- `nonzero()`-selected
# positives, a python loop over the gts, per-level `loss_single` -- written independently of the fixed-shape masked form the
# product uses (sm3det_amd/gfl_losses.py, sm3det_amd/gfl_head.py); tests/test_gfl_loss_cpu.py compares the two.  Sources
# restated (mmdet 2.25, not vendored: PARITY UNPINNED): core/bbox/assigners/atss_assigner.py:47-201,
# models/dense_heads/gfl_head.py:16-50 (Integral), :233-343 (loss_single), :345-420 (loss), :575-648 (get_targets /
# _get_target_single), models/losses/gfocal_loss.py:12-52,95-118, models/losses/iou_loss.py:120-135,
# core/bbox/transforms.py (distance2bbox / bbox2distance), core/bbox/iou_calculators/iou2d_calculator.py (bbox_overlaps).
import torch  # noqa: E402


def _iou_pairwise(b1, b2, eps=1e-6):
    out = torch.zeros(b1.shape[0], b2.shape[0])
    for i in range(b1.shape[0]):
        for j in range(b2.shape[0]):
            out[i, j] = _iou_one(b1[i], b2[j], eps)
    return out


def _iou_one(a, b, eps=1e-6, giou=False):
    a1 = (a[2] - a[0]) * (a[3] - a[1])
    a2 = (b[2] - b[0]) * (b[3] - b[1])
    w = (torch.min(a[2], b[2]) - torch.max(a[0], b[0])).clamp(min=0)
    h = (torch.min(a[3], b[3]) - torch.max(a[1], b[1])).clamp(min=0)
    ov = w * h
    un = torch.max(a1 + a2 - ov, torch.tensor(eps))
    iou = ov / un
    if not giou:
        return iou
    ew = (torch.max(a[2], b[2]) - torch.min(a[0], b[0])).clamp(min=0)
    eh = (torch.max(a[3], b[3]) - torch.min(a[1], b[1])).clamp(min=0)
    ea = torch.max(ew * eh, torch.tensor(eps))
    return iou - (ea - un) / ea


def iou_matrix(b1, b2, eps=1e-6):
    """vectorised form of the pair loop above (the loop is kept for the small cross-check in the test)"""
    lt = torch.max(b1[:, None, :2], b2[None, :, :2])
    rb = torch.min(b1[:, None, 2:], b2[None, :, 2:])
    wh = (rb - lt).clamp(min=0)
    ov = wh[..., 0] * wh[..., 1]
    a1 = (b1[:, 2] - b1[:, 0]) * (b1[:, 3] - b1[:, 1])
    a2 = (b2[:, 2] - b2[:, 0]) * (b2[:, 3] - b2[:, 1])
    return ov / torch.max(a1[:, None] + a2[None, :] - ov, torch.tensor(eps))


def atss_assign(bboxes, num_level_bboxes, gt_bboxes, gt_labels, topk=9):
    """atss_assigner.py:47-201, statement order kept (flattened candidate indices, per-gt loop)"""
    INF = 100000000
    bboxes = bboxes[:, :4]
    num_gt, num_bboxes = gt_bboxes.size(0), bboxes.size(0)
    overlaps = iou_matrix(bboxes, gt_bboxes)
    assigned_gt_inds = overlaps.new_full((num_bboxes,), 0, dtype=torch.long)
    if num_gt == 0 or num_bboxes == 0:
        max_overlaps = overlaps.new_zeros((num_bboxes,))
        labels = None if gt_labels is None else overlaps.new_full((num_bboxes,), -1, dtype=torch.long)
        return assigned_gt_inds, max_overlaps, labels
    gt_cx = (gt_bboxes[:, 0] + gt_bboxes[:, 2]) / 2.0
    gt_cy = (gt_bboxes[:, 1] + gt_bboxes[:, 3]) / 2.0
    gt_points = torch.stack((gt_cx, gt_cy), dim=1)
    bboxes_cx = (bboxes[:, 0] + bboxes[:, 2]) / 2.0
    bboxes_cy = (bboxes[:, 1] + bboxes[:, 3]) / 2.0
    bboxes_points = torch.stack((bboxes_cx, bboxes_cy), dim=1)
    distances = (bboxes_points[:, None, :] - gt_points[None, :, :]).pow(2).sum(-1).sqrt()
    candidate_idxs = []
    start_idx = 0
    for bboxes_per_level in num_level_bboxes:
        end_idx = start_idx + bboxes_per_level
        distances_per_level = distances[start_idx:end_idx, :]
        selectable_k = min(topk, bboxes_per_level)
        _, topk_idxs_per_level = distances_per_level.topk(selectable_k, dim=0, largest=False)
        candidate_idxs.append(topk_idxs_per_level + start_idx)
        start_idx = end_idx
    candidate_idxs = torch.cat(candidate_idxs, dim=0)
    candidate_overlaps = overlaps[candidate_idxs, torch.arange(num_gt)]
    overlaps_thr_per_gt = candidate_overlaps.mean(0) + candidate_overlaps.std(0)
    is_pos = candidate_overlaps >= overlaps_thr_per_gt[None, :]
    for gt_idx in range(num_gt):
        candidate_idxs[:, gt_idx] += gt_idx * num_bboxes
    ep_bboxes_cx = bboxes_cx.view(1, -1).expand(num_gt, num_bboxes).contiguous().view(-1)
    ep_bboxes_cy = bboxes_cy.view(1, -1).expand(num_gt, num_bboxes).contiguous().view(-1)
    candidate_idxs = candidate_idxs.view(-1)
    l_ = ep_bboxes_cx[candidate_idxs].view(-1, num_gt) - gt_bboxes[:, 0]
    t_ = ep_bboxes_cy[candidate_idxs].view(-1, num_gt) - gt_bboxes[:, 1]
    r_ = gt_bboxes[:, 2] - ep_bboxes_cx[candidate_idxs].view(-1, num_gt)
    b_ = gt_bboxes[:, 3] - ep_bboxes_cy[candidate_idxs].view(-1, num_gt)
    is_in_gts = torch.stack([l_, t_, r_, b_], dim=1).min(dim=1)[0] > 0.01
    is_pos = is_pos & is_in_gts
    overlaps_inf = torch.full_like(overlaps, -INF).t().contiguous().view(-1)
    index = candidate_idxs.view(-1)[is_pos.view(-1)]
    overlaps_inf[index] = overlaps.t().contiguous().view(-1)[index]
    overlaps_inf = overlaps_inf.view(num_gt, -1).t()
    max_overlaps, argmax_overlaps = overlaps_inf.max(dim=1)
    assigned_gt_inds[max_overlaps != -INF] = argmax_overlaps[max_overlaps != -INF] + 1
    labels = None
    if gt_labels is not None:
        labels = assigned_gt_inds.new_full((num_bboxes,), -1)
        pos_inds = torch.nonzero(assigned_gt_inds > 0, as_tuple=False).squeeze(1)
        if pos_inds.numel() > 0:
            labels[pos_inds] = gt_labels[assigned_gt_inds[pos_inds] - 1]
    return assigned_gt_inds, max_overlaps, labels


def get_target_single(flat_anchors, num_level_anchors, gt_bboxes, gt_labels, num_classes, pos_weight=-1, topk=9):
    """gfl_head.py:_get_target_single with every anchor inside the image (allowed_border -1, full valid flags)"""
    gt_inds, _, _ = atss_assign(flat_anchors, num_level_anchors, gt_bboxes, gt_labels, topk)
    pos_inds = torch.nonzero(gt_inds > 0, as_tuple=False).squeeze(-1).unique()
    neg_inds = torch.nonzero(gt_inds == 0, as_tuple=False).squeeze(-1).unique()
    n = flat_anchors.shape[0]
    bbox_targets = torch.zeros_like(flat_anchors)
    labels = flat_anchors.new_full((n,), num_classes, dtype=torch.long)
    label_weights = flat_anchors.new_zeros(n)
    if len(pos_inds) > 0:
        bbox_targets[pos_inds, :] = gt_bboxes[gt_inds[pos_inds] - 1]
        labels[pos_inds] = gt_labels[gt_inds[pos_inds] - 1]
        label_weights[pos_inds] = 1.0 if pos_weight <= 0 else pos_weight
    if len(neg_inds) > 0:
        label_weights[neg_inds] = 1.0
    return labels, label_weights, bbox_targets, pos_inds


def quality_focal_loss(pred, label, score, beta=2.0):
    pred_sigmoid = pred.sigmoid()
    zerolabel = pred_sigmoid.new_zeros(pred.shape)
    loss = F.binary_cross_entropy_with_logits(pred, zerolabel, reduction='none') * pred_sigmoid.pow(beta)
    bg = pred.size(1)
    pos = ((label >= 0) & (label < bg)).nonzero().squeeze(1)
    pos_label = label[pos].long()
    sf = score[pos] - pred_sigmoid[pos, pos_label]
    loss[pos, pos_label] = F.binary_cross_entropy_with_logits(pred[pos, pos_label], score[pos], reduction='none') * sf.abs().pow(beta)
    return loss.sum(dim=1)


def distribution_focal_loss(pred, label):
    dis_left = label.long()
    dis_right = dis_left + 1
    return (F.cross_entropy(pred, dis_left, reduction='none') * (dis_right.float() - label) +
            F.cross_entropy(pred, dis_right, reduction='none') * (label - dis_left.float()))


def gfl_loss(cls_scores, bbox_preds, lvl_anchors, strides, gt_bboxes, gt_labels, num_classes, reg_max=16, beta=2.0,
             w_cls=1.0, w_dfl=0.25, w_box=2.0, topk=9):
    """GFLHead.loss for images whose pad_shape covers every feature map: dict of per-level lists"""
    num_level = [a.shape[0] for a in lvl_anchors]
    flat = torch.cat(lvl_anchors)
    B = cls_scores[0].shape[0]
    tg = [get_target_single(flat, num_level, gt_bboxes[i], gt_labels[i], num_classes, -1, topk) for i in range(B)]
    num_total_pos = sum(max(int(t[3].numel()), 1) for t in tg)  # get_targets: max(inds.numel(), 1) per image
    num_total_samples = max(float(num_total_pos), 1.0)
    proj = torch.linspace(0, reg_max, reg_max + 1)
    losses_cls, losses_bbox, losses_dfl, avg = [], [], [], []
    start = 0
    for lv, stride in enumerate(strides):
        nl = num_level[lv]
        anchors = flat[start:start + nl].repeat(B, 1)
        cls_score = cls_scores[lv].permute(0, 2, 3, 1).reshape(-1, num_classes)
        bbox_pred = bbox_preds[lv].permute(0, 2, 3, 1).reshape(-1, 4 * (reg_max + 1))
        labels = torch.cat([t[0][start:start + nl] for t in tg])
        label_weights = torch.cat([t[1][start:start + nl] for t in tg])
        bbox_targets = torch.cat([t[2][start:start + nl] for t in tg])
        start += nl
        pos_inds = ((labels >= 0) & (labels < num_classes)).nonzero().squeeze(1)
        score = label_weights.new_zeros(labels.shape)
        if len(pos_inds) > 0:
            pos_bbox_targets = bbox_targets[pos_inds]
            pos_bbox_pred = bbox_pred[pos_inds]
            pa = anchors[pos_inds]
            centers = torch.stack(((pa[:, 0] + pa[:, 2]) / 2, (pa[:, 1] + pa[:, 3]) / 2), dim=-1) / stride
            weight_targets = cls_score.detach().sigmoid().max(dim=1)[0][pos_inds]
            corners = F.linear(F.softmax(pos_bbox_pred.reshape(-1, reg_max + 1), dim=1), proj[None]).reshape(-1, 4)
            dec = torch.stack((centers[:, 0] - corners[:, 0], centers[:, 1] - corners[:, 1],
                               centers[:, 0] + corners[:, 2], centers[:, 1] + corners[:, 3]), -1)
            dec_t = pos_bbox_targets / stride
            score[pos_inds] = torch.stack([_iou_one(a, b) for a, b in zip(dec.detach(), dec_t)])
            pred_corners = pos_bbox_pred.reshape(-1, reg_max + 1)
            tc = torch.stack((centers[:, 0] - dec_t[:, 0], centers[:, 1] - dec_t[:, 1],
                              dec_t[:, 2] - centers[:, 0], dec_t[:, 3] - centers[:, 1]), -1).clamp(min=0, max=reg_max - 0.1)
            giou = torch.stack([_iou_one(a, b, 1e-7, giou=True) for a, b in zip(dec, dec_t)])
            loss_bbox = ((1 - giou) * weight_targets).sum() / 1.0 * w_box
            loss_dfl = (distribution_focal_loss(pred_corners, tc.reshape(-1)) *
                        weight_targets[:, None].expand(-1, 4).reshape(-1)).sum() / 4.0 * w_dfl
        else:
            loss_bbox = bbox_pred.sum() * 0
            loss_dfl = bbox_pred.sum() * 0
            weight_targets = bbox_pred.new_tensor(0)
        loss_cls = (quality_focal_loss(cls_score, labels, score, beta) * label_weights).sum() / num_total_samples * w_cls
        losses_cls.append(loss_cls)
        losses_bbox.append(loss_bbox)
        losses_dfl.append(loss_dfl)
        avg.append(weight_targets.sum())
    avg_factor = max(float(sum(avg)), 1.0)
    return dict(loss_cls=losses_cls, loss_bbox=[v / avg_factor for v in losses_bbox],
                loss_dfl=[v / avg_factor for v in losses_dfl])
