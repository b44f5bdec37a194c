"""The loss side of the SAR branch's ``GFLHead`` (``local_configs/main_SM3Det.py:29-48,145-149``): ``ATSSAssigner(topk=9)``,
``QualityFocalLoss(beta=2)``, ``DistributionFocalLoss``, ``GIoULoss``, the ``Integral`` layer and the distance-point coder.

All of this is **mmdet 2.x code that the reference does not vendor** (``mmdet/core/bbox/assigners/atss_assigner.py``,
``mmdet/models/losses/{gfocal_loss,iou_loss}.py``, ``mmdet/models/dense_heads/gfl_head.py``, ``mmdet/core/bbox/transforms.py``):
the arithmetic below restates the published implementation -- **parity unpinned** (nothing under /root/reference states
it; ``oracle/gfl_oracle.py`` holds a second, independently written restatement in mmdet's own indexing form and
``tests/test_gfl_loss_cpu.py`` compares the two).  SURVEY.md 7 ("hard parts") / VERDICT r03 item 7: pure PyTorch by design --
the SAR targets are O(21 824 anchors x 8 gts) per image, nowhere near the hot path's cost.

MI355X-side design choice: mmdet's versions select the positives with ``nonzero()`` (a device->host sync per level and
image, dynamic shapes) and loop ``for gt_idx in range(num_gt)`` on the host.  Here every tensor keeps a FIXED shape and
the positives are a mask / a weight of zero -- the same sums term for term (a zero-weight term contributes exactly 0), no
synchronisation, so the whole detector step stays capturable in a hipGraph.
"""
import torch
import torch.nn.functional as F

INF = 100000000.0


def bbox_overlaps(b1, b2, mode='iou', is_aligned=False, eps=1e-6):
    """mmdet ``bbox_overlaps`` for (x1, y1, x2, y2) boxes: iou / giou, pairwise (n, k) or aligned (n,)."""
    a1 = (b1[..., 2] - b1[..., 0]) * (b1[..., 3] - b1[..., 1])
    a2 = (b2[..., 2] - b2[..., 0]) * (b2[..., 3] - b2[..., 1])
    if is_aligned:
        lt = torch.max(b1[..., :2], b2[..., :2])
        rb = torch.min(b1[..., 2:], b2[..., 2:])
        wh = (rb - lt).clamp(min=0)
        overlap = wh[..., 0] * wh[..., 1]
        union = a1 + a2 - overlap
        if mode == 'giou':
            elt = torch.min(b1[..., :2], b2[..., :2])
            erb = torch.max(b1[..., 2:], b2[..., 2:])
    else:
        lt = torch.max(b1[:, None, :2], b2[None, :, :2])
        rb = torch.min(b1[:, None, 2:], b2[None, :, 2:])
        wh = (rb - lt).clamp(min=0)
        overlap = wh[..., 0] * wh[..., 1]
        union = a1[:, None] + a2[None, :] - overlap
        if mode == 'giou':
            elt = torch.min(b1[:, None, :2], b2[None, :, :2])
            erb = torch.max(b1[:, None, 2:], b2[None, :, 2:])
    union = union.clamp(min=eps)  # max(union, eps) without a host scalar upload (hipGraph capture)
    ious = overlap / union
    if mode == 'iou':
        return ious
    ewh = (erb - elt).clamp(min=0)
    earea = (ewh[..., 0] * ewh[..., 1]).clamp(min=eps)
    return ious - (earea - union) / earea


def atss_assign(bboxes, num_level_bboxes, gt_bboxes, gt_labels=None, topk=9, valid=None):
    """``ATSSAssigner.assign`` (mmdet 2.25 ``atss_assigner.py:47-201``; Zhang et al., ATSS, CVPR 2020) without host loops:

    1. IoU of every anchor with every gt; centre distance of every anchor to every gt;
    2. per pyramid level the ``topk`` anchors closest to each gt are its candidates (k x levels per gt);
    3. per gt: threshold = mean + std (unbiased) of its candidates' IoUs; a candidate is positive if its IoU >= threshold
       and its centre lies inside the gt (min side distance > 0.01);
    4. an anchor positive for several gts goes to the one with the highest IoU.

    bboxes (n, 4) level-major; num_level_bboxes: python ints; gt_bboxes (k, 4); valid (n,) bool or None: anchors outside
    the padded image (mmdet removes them before assigning; here they get an infinite distance, never become positives, and
    are left out of the mean + std threshold -- the same assignment as the compacted form).  Returns (gt_inds (n,) long: 0 negative / i + 1, max_overlaps (n,),
    labels (n,) long or None: -1 where unassigned)."""
    n, k = bboxes.size(0), gt_bboxes.size(0)
    bboxes = bboxes[:, :4]
    if k == 0 or n == 0:
        gt_inds = bboxes.new_zeros(n, dtype=torch.long)
        return gt_inds, bboxes.new_zeros(n), (None if gt_labels is None else gt_inds.new_full((n,), -1))
    overlaps = bbox_overlaps(bboxes, gt_bboxes)  # (n, k)
    gt_pts = torch.stack(((gt_bboxes[:, 0] + gt_bboxes[:, 2]) / 2.0, (gt_bboxes[:, 1] + gt_bboxes[:, 3]) / 2.0), dim=1)
    cx, cy = (bboxes[:, 0] + bboxes[:, 2]) / 2.0, (bboxes[:, 1] + bboxes[:, 3]) / 2.0
    pts = torch.stack((cx, cy), dim=1)
    dist = (pts[:, None, :] - gt_pts[None, :, :]).pow(2).sum(-1).sqrt()
    if valid is not None:
        dist = torch.where(valid[:, None], dist, dist.new_full((), float('inf')))
    cand, start = [], 0
    for nl in num_level_bboxes:
        sel = min(topk, nl)
        _, idx = dist[start:start + nl].topk(sel, dim=0, largest=False)
        cand.append(idx + start)
        start += nl
    cand = torch.cat(cand, dim=0)  # (c, k)
    cand_ov = overlaps.gather(0, cand)
    if valid is None:
        thr = cand_ov.mean(0) + cand_ov.std(0)
    else:
        # mmdet removes the anchors outside the padded image BEFORE assigning (and shrinks topk to the valid anchors a
        # level holds), so only valid candidates enter a gt's mean + std threshold: masked moments (unbiased std, as
        # torch.std) over the candidates that are valid -- equal to the compacted form whatever a level's valid count is
        vc = valid[cand].to(cand_ov.dtype)
        cnt = vc.sum(0)
        mean = (cand_ov * vc).sum(0) / cnt
        thr = mean + (((cand_ov - mean[None, :]) ** 2 * vc).sum(0) / (cnt - 1)).sqrt()
    is_pos = cand_ov >= thr[None, :]
    ccx, ccy = cx[cand], cy[cand]  # (c, k)
    side = torch.stack((ccx - gt_bboxes[None, :, 0], ccy - gt_bboxes[None, :, 1],
                        gt_bboxes[None, :, 2] - ccx, gt_bboxes[None, :, 3] - ccy), dim=0).min(dim=0)[0]
    is_pos = is_pos & (side > 0.01)
    if valid is not None:
        is_pos = is_pos & valid[cand]
    pos_mask = torch.zeros(n, k, dtype=torch.bool, device=bboxes.device)
    pos_mask.scatter_(0, cand, is_pos)  # a gt's candidates are distinct anchors: no write conflicts inside a column
    ov_inf = torch.where(pos_mask, overlaps, overlaps.new_full((), -INF))
    max_ov, arg = ov_inf.max(dim=1)
    has = max_ov != -INF
    gt_inds = torch.where(has, arg + 1, torch.zeros_like(arg))
    labels = None
    if gt_labels is not None:
        labels = torch.where(has, gt_labels.long()[arg], gt_inds.new_full((), -1))
    return gt_inds, max_ov, labels


def integral(x, reg_max):
    """``Integral.forward`` (gfl_head.py:16-50): expectation of the softmax distribution over {0..reg_max} per box side"""
    p = F.softmax(x.reshape(-1, reg_max + 1), dim=1)
    proj = torch.linspace(0, reg_max, reg_max + 1, device=x.device, dtype=p.dtype)
    return F.linear(p, proj[None, :]).reshape(-1, 4)


def distance2bbox(points, distance):
    """``DistancePointBBoxCoder.decode`` without max_shape (transforms.py distance2bbox)"""
    return torch.stack((points[..., 0] - distance[..., 0], points[..., 1] - distance[..., 1],
                        points[..., 0] + distance[..., 2], points[..., 1] + distance[..., 3]), dim=-1)


def bbox2distance(points, bbox, max_dis=None, eps=0.1):
    """``DistancePointBBoxCoder.encode`` (transforms.py bbox2distance): side distances clamped to [0, max_dis - eps]"""
    d = torch.stack((points[..., 0] - bbox[..., 0], points[..., 1] - bbox[..., 1],
                     bbox[..., 2] - points[..., 0], bbox[..., 3] - points[..., 1]), dim=-1)
    if max_dis is not None:
        d = d.clamp(min=0, max=max_dis - eps)
    return d


def quality_focal_loss(pred, label, score, pos, beta=2.0):
    """``quality_focal_loss`` (gfocal_loss.py:12-52) per anchor (sum over classes), mask form: pred (n, C) logits;
    label (n,) class of the positives; score (n,) IoU quality target; pos (n,) bool."""
    sig = pred.sigmoid()
    loss = F.binary_cross_entropy_with_logits(pred, torch.zeros_like(pred), reduction='none') * sig.pow(beta)
    onehot = F.one_hot(label.clamp(min=0, max=pred.size(1) - 1), pred.size(1)).bool() & pos[:, None]
    tgt = score[:, None].expand_as(pred)
    pos_loss = F.binary_cross_entropy_with_logits(pred, tgt, reduction='none') * (tgt - sig).abs().pow(beta)
    return torch.where(onehot, pos_loss, loss).sum(dim=1)


def distribution_focal_loss(pred, label):
    """``distribution_focal_loss`` (gfocal_loss.py:95-118): pred (m, reg_max + 1) logits, label (m,) in [0, reg_max)"""
    dl = label.long()
    dr = dl + 1
    wl = dr.float() - label
    wr = label - dl.float()
    return F.cross_entropy(pred, dl, reduction='none') * wl + F.cross_entropy(pred, dr, reduction='none') * wr


def giou_loss(pred, target, eps=1e-7):
    """``giou_loss`` (iou_loss.py:120-135): 1 - GIoU of aligned boxes"""
    return 1.0 - bbox_overlaps(pred, target, mode='giou', is_aligned=True, eps=eps)
