"""CPU oracle of the Oriented-RPN conv tower and proposal glue -- TEST INFRASTRUCTURE ONLY.

torch-CPU restatements, each citing the reference lines it follows:
* ``rpn_forward_single``  -- RotatedRPNHead.forward_single, mmrotate/models/dense_heads/rotated_rpn_head.py:43-50 with
  the layers of OrientedRPNHead._init_layers (oriented_rpn_head.py:18-24);
* ``delta2bbox``          -- mmrotate/core/bbox/coder/delta_midpointoffset_rbbox_coder.py:150-238;
* ``poly2obb_le90`` / ``norm_angle`` / ``obb2xyxy_le90`` -- mmrotate/core/bbox/transforms.py:301-331, :850-867, :685-702;
* ``get_bboxes_single``   -- OrientedRPNHead._get_bboxes_single, oriented_rpn_head.py:189-281, with mmcv's
  ``batched_nms`` (mmcv/mmcv/ops/nms.py:264-382) over the plain-C NMS oracle (oracle/ops_oracle.py).

Pinned: tests/test_oracle_rpn.py compares the box functions with the reference functions imported from /root/reference
(oracle/ref_rpn.py) when present, and with fixtures generated from those (tests/golden/make_golden_rpn.py);
tests/test_oracle_heads_live.py runs the reference's own ``OrientedRPNHead._init_layers`` / ``forward_single`` /
``_get_bboxes_single`` live (with the reference's own ``batched_nms`` over its compiled CPU ``nms``): ``rpn_forward_single``
and ``get_bboxes_single`` reproduce them bit for bit."""
import numpy as np
import torch
import torch.nn.functional as F


def rpn_forward_single(x, p):
    x = F.conv2d(x, p['rpn_conv.weight'], p['rpn_conv.bias'], padding=1)
    x = F.relu(x)
    return F.conv2d(x, p['rpn_cls.weight'], p['rpn_cls.bias']), F.conv2d(x, p['rpn_reg.weight'], p['rpn_reg.bias'])


def norm_angle(angle, angle_range):
    assert angle_range == 'le90'
    return (angle + np.pi / 2) % np.pi - np.pi / 2


def poly2obb_le90(polys):
    polys = torch.reshape(polys, [-1, 8])
    pt1, pt2, pt3, pt4 = polys[..., :8].chunk(4, 1)
    edge1 = torch.sqrt(torch.pow(pt1[..., 0] - pt2[..., 0], 2) + torch.pow(pt1[..., 1] - pt2[..., 1], 2))
    edge2 = torch.sqrt(torch.pow(pt2[..., 0] - pt3[..., 0], 2) + torch.pow(pt2[..., 1] - pt3[..., 1], 2))
    angles1 = torch.atan2((pt2[..., 1] - pt1[..., 1]), (pt2[..., 0] - pt1[..., 0]))
    angles2 = torch.atan2((pt4[..., 1] - pt1[..., 1]), (pt4[..., 0] - pt1[..., 0]))
    angles = polys.new_zeros(polys.shape[0])
    angles[edge1 > edge2] = angles1[edge1 > edge2]
    angles[edge1 <= edge2] = angles2[edge1 <= edge2]
    angles = norm_angle(angles, 'le90')
    x_ctr = (pt1[..., 0] + pt3[..., 0]) / 2.0
    y_ctr = (pt1[..., 1] + pt3[..., 1]) / 2.0
    edges = torch.stack([edge1, edge2], dim=1)
    width, _ = torch.max(edges, 1)
    height, _ = torch.min(edges, 1)
    return torch.stack([x_ctr, y_ctr, width, height, angles], 1)


def obb2xyxy_le90(obboxes):
    center, w, h, theta = torch.split(obboxes, [2, 1, 1, 1], dim=-1)
    Cos, Sin = torch.cos(theta), torch.sin(theta)
    x_bias = torch.abs(w / 2 * Cos) + torch.abs(h / 2 * Sin)
    y_bias = torch.abs(w / 2 * Sin) + torch.abs(h / 2 * Cos)
    bias = torch.cat([x_bias, y_bias], dim=-1)
    return torch.cat([center - bias, center + bias], dim=-1)


def delta2bbox(rois, deltas, means=(0.,) * 6, stds=(1.,) * 6, wh_ratio_clip=16 / 1000):
    means = deltas.new_tensor(means).repeat(1, deltas.size(1) // 6)
    stds = deltas.new_tensor(stds).repeat(1, deltas.size(1) // 6)
    dd = deltas * stds + means
    dx, dy, dw, dh, da, db = (dd[:, k::6] for k in range(6))
    max_ratio = np.abs(np.log(wh_ratio_clip))
    dw = dw.clamp(min=-max_ratio, max=max_ratio)
    dh = dh.clamp(min=-max_ratio, max=max_ratio)
    px = ((rois[:, 0] + rois[:, 2]) * 0.5).unsqueeze(1).expand_as(dx)
    py = ((rois[:, 1] + rois[:, 3]) * 0.5).unsqueeze(1).expand_as(dy)
    pw = (rois[:, 2] - rois[:, 0]).unsqueeze(1).expand_as(dw)
    ph = (rois[:, 3] - rois[:, 1]).unsqueeze(1).expand_as(dh)
    gw, gh = pw * dw.exp(), ph * dh.exp()
    gx, gy = px + pw * dx, py + ph * dy
    x1, y1, x2, y2 = gx - gw * 0.5, gy - gh * 0.5, gx + gw * 0.5, gy + gh * 0.5
    da = da.clamp(min=-0.5, max=0.5)
    db = db.clamp(min=-0.5, max=0.5)
    ga, _ga, gb, _gb = gx + da * gw, gx - da * gw, gy + db * gh, gy - db * gh
    polys = torch.stack([ga, y1, x2, gb, _ga, y2, x1, _gb], dim=-1)
    center = torch.stack([gx, gy, gx, gy, gx, gy, gx, gy], dim=-1)
    cp = polys - center
    diag_len = torch.sqrt(cp[..., 0::2] * cp[..., 0::2] + cp[..., 1::2] * cp[..., 1::2])
    max_diag_len, _ = torch.max(diag_len, dim=-1, keepdim=True)
    cp = cp * (max_diag_len / diag_len).repeat_interleave(2, dim=-1)
    return poly2obb_le90(cp + center)


def obb2poly_le90(rboxes):
    """transforms.py:474-499"""
    N = rboxes.shape[0]
    if N == 0:
        return rboxes.new_zeros((rboxes.size(0), 8))
    x_ctr, y_ctr, width, height, angle = (rboxes.select(1, k) for k in range(5))
    tl_x, tl_y, br_x, br_y = -width * 0.5, -height * 0.5, width * 0.5, height * 0.5
    rects = torch.stack([tl_x, br_x, br_x, tl_x, tl_y, tl_y, br_y, br_y], dim=0).reshape(2, 4, N).permute(2, 0, 1)
    sin, cos = torch.sin(angle), torch.cos(angle)
    M = torch.stack([cos, -sin, sin, cos], dim=0).reshape(2, 2, N).permute(2, 0, 1)
    polys = M.matmul(rects).permute(2, 1, 0).reshape(-1, N).transpose(1, 0)
    polys[:, ::2] += x_ctr.unsqueeze(1)
    polys[:, 1::2] += y_ctr.unsqueeze(1)
    return polys.contiguous()


def midpoint_bbox2delta(proposals, gt, means=(0.,) * 6, stds=(1.,) * 6):
    """delta_midpointoffset_rbbox_coder.py:87-148 (version le90)"""
    proposals, gt = proposals.float(), gt.float()
    px = (proposals[..., 0] + proposals[..., 2]) * 0.5
    py = (proposals[..., 1] + proposals[..., 3]) * 0.5
    pw = proposals[..., 2] - proposals[..., 0]
    ph = proposals[..., 3] - proposals[..., 1]
    hbb, poly = obb2xyxy_le90(gt), obb2poly_le90(gt)
    gx = (hbb[..., 0] + hbb[..., 2]) * 0.5
    gy = (hbb[..., 1] + hbb[..., 3]) * 0.5
    gw = hbb[..., 2] - hbb[..., 0]
    gh = hbb[..., 3] - hbb[..., 1]
    x_coor, y_coor = poly[:, 0::2], poly[:, 1::2]
    y_min, _ = torch.min(y_coor, dim=1, keepdim=True)
    x_max, _ = torch.max(x_coor, dim=1, keepdim=True)
    _x_coor = x_coor.clone()
    _x_coor[torch.abs(y_coor - y_min) > 0.1] = -1000
    ga, _ = torch.max(_x_coor, dim=1)
    _y_coor = y_coor.clone()
    _y_coor[torch.abs(x_coor - x_max) > 0.1] = -1000
    gb, _ = torch.max(_y_coor, dim=1)
    deltas = torch.stack([(gx - px) / pw, (gy - py) / ph, torch.log(gw / pw), torch.log(gh / ph), (ga - gx) / gw,
                          (gb - gy) / gh], dim=-1)
    return deltas.sub_(deltas.new_tensor(means).unsqueeze(0)).div_(deltas.new_tensor(stds).unsqueeze(0))


def xywha_bbox2delta(proposals, gt, means, stds, norm_factor=None, edge_swap=False, proj_xy=False):
    """delta_xywha_rbbox_coder.py:112-176 (angle_range le90)"""
    px, py, pw, ph, pa = proposals.float().unbind(dim=-1)
    gx, gy, gw, gh, ga = gt.float().unbind(dim=-1)
    if proj_xy:
        dx = (torch.cos(pa) * (gx - px) + torch.sin(pa) * (gy - py)) / pw
        dy = (-torch.sin(pa) * (gx - px) + torch.cos(pa) * (gy - py)) / ph
    else:
        dx, dy = (gx - px) / pw, (gy - py) / ph
    if edge_swap:
        dtheta1 = norm_angle(ga - pa, 'le90')
        dtheta2 = norm_angle(ga - pa + np.pi / 2, 'le90')
        first = torch.abs(dtheta1) < torch.abs(dtheta2)
        da = torch.where(first, dtheta1, dtheta2)
        dw = torch.log(torch.where(first, gw, gh) / pw)
        dh = torch.log(torch.where(first, gh, gw) / ph)
    else:
        da = norm_angle(ga - pa, 'le90')
        dw, dh = torch.log(gw / pw), torch.log(gh / ph)
    if norm_factor:
        da /= norm_factor * np.pi
    deltas = torch.stack([dx, dy, dw, dh, da], dim=-1)
    return deltas.sub_(deltas.new_tensor(means).unsqueeze(0)).div_(deltas.new_tensor(stds).unsqueeze(0))


def xywha_delta2bbox(rois, deltas, means, stds, max_shape=None, wh_ratio_clip=16 / 1000, norm_factor=None,
                     edge_swap=False, proj_xy=False):
    """delta_xywha_rbbox_coder.py:180-283 (angle_range le90, add_ctr_clamp False, (N,5) deltas)"""
    dd = deltas * deltas.new_tensor(stds).view(1, -1) + deltas.new_tensor(means).view(1, -1)
    dx, dy, dw, dh, da = (dd[:, k::5] for k in range(5))
    if norm_factor:
        da = da * (norm_factor * np.pi)
    px, py, pw, ph, pa = (rois[:, k].unsqueeze(1).expand_as(dx) for k in range(5))
    dx_width, dy_height = pw * dx, ph * dy
    max_ratio = np.abs(np.log(wh_ratio_clip))
    dw = dw.clamp(min=-max_ratio, max=max_ratio)
    dh = dh.clamp(min=-max_ratio, max=max_ratio)
    gw, gh = pw * dw.exp(), ph * dh.exp()
    if proj_xy:
        gx = dx * pw * torch.cos(pa) - dy * ph * torch.sin(pa) + px
        gy = dx * pw * torch.sin(pa) + dy * ph * torch.cos(pa) + py
    else:
        gx, gy = px + dx_width, py + dy_height
    ga = norm_angle(pa + da, 'le90')
    if max_shape is not None:
        gx = gx.clamp(min=0, max=max_shape[1] - 1)
        gy = gy.clamp(min=0, max=max_shape[0] - 1)
    if edge_swap:
        w_regular = torch.where(gw > gh, gw, gh)
        h_regular = torch.where(gw > gh, gh, gw)
        theta_regular = norm_angle(torch.where(gw > gh, ga, ga + np.pi / 2), 'le90')
        return torch.stack([gx, gy, w_regular, h_regular, theta_regular], dim=-1).view_as(deltas)
    return torch.stack([gx, gy, gw, gh, ga], dim=-1).view(deltas.size())


def batched_nms(boxes, scores, idxs, iou_threshold):
    """mmcv batched_nms (type 'nms', below split_thr): offset trick + the plain-C NMS oracle; returns keep"""
    from oracle import ops_oracle as OO
    max_coordinate = boxes.max()
    offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
    keep = OO.nms((boxes + offsets[:, None]).numpy(), scores.numpy(), float(iou_threshold), 0)
    return torch.as_tensor(np.asarray(keep), dtype=torch.long)


def get_bboxes_single(cls_scores, bbox_preds, mlvl_anchors, cfg, means, stds):
    level_ids, mlvl_scores, mlvl_bbox_preds, mlvl_valid_anchors = [], [], [], []
    for idx in range(len(cls_scores)):
        scores = cls_scores[idx].permute(1, 2, 0).reshape(-1).sigmoid()
        pred = bbox_preds[idx].permute(1, 2, 0).reshape(-1, 6)
        anchors = mlvl_anchors[idx]
        if cfg['nms_pre'] > 0 and scores.shape[0] > cfg['nms_pre']:
            ranked_scores, rank_inds = scores.sort(descending=True, stable=True)
            topk_inds = rank_inds[:cfg['nms_pre']]
            scores = ranked_scores[:cfg['nms_pre']]
            pred, anchors = pred[topk_inds, :], anchors[topk_inds, :]
        mlvl_scores.append(scores)
        mlvl_bbox_preds.append(pred)
        mlvl_valid_anchors.append(anchors)
        level_ids.append(scores.new_full((scores.size(0),), idx, dtype=torch.long))
    scores, anchors = torch.cat(mlvl_scores), torch.cat(mlvl_valid_anchors)
    proposals = delta2bbox(anchors, torch.cat(mlvl_bbox_preds), means, stds)
    ids = torch.cat(level_ids)
    if cfg.get('min_bbox_size', 0) > 0:
        valid = (proposals[:, 2] >= cfg['min_bbox_size']) & (proposals[:, 3] >= cfg['min_bbox_size'])
        proposals, scores, ids = proposals[valid], scores[valid], ids[valid]
    if proposals.numel() == 0:
        return proposals.new_zeros(0, 5)
    keep = batched_nms(obb2xyxy_le90(proposals), scores, ids, cfg['nms']['iou_threshold'])
    return torch.cat([proposals, scores[:, None]], dim=1)[keep][:cfg['max_per_img']]
