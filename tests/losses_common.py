"""Seeded cases shared by the CPU test that pins oracle/loss_oracle.py to the live reference heads and the GPU test that
compares the HIP loss kernels with that oracle."""
import numpy as np
import torch

from oracle import assign_oracle, loss_oracle as LO, rpn_oracle

RPN_MEANS, RPN_STDS = (0.0,) * 6, (1.0, 1.0, 1.0, 1.0, 0.5, 0.5)          # main_SM3Det.py:59-63
RCNN_MEANS, RCNN_STDS = (0.0,) * 5, (0.1, 0.1, 0.2, 0.2, 0.1)             # main_SM3Det.py:84-92
RPN_ASSIGN = dict(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, match_low_quality=True)   # :166-172
RCNN_ASSIGN = dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False)  # :184-191


def cpu_grid_anchors(sizes, strides, scales=(8,), ratios=(0.5, 1.0, 2.0)):
    from sm3det_amd.rpn_head import grid_anchors
    return grid_anchors(sizes, strides, scales, ratios, device='cpu')


def oriented_gts(k, extent, seed, lo=8.0, hi=48.0):
    g = torch.Generator().manual_seed(seed)
    b = torch.zeros(k, 5)
    b[:, :2] = torch.rand(k, 2, generator=g) * (extent * 0.7) + extent * 0.15
    b[:, 2] = torch.rand(k, generator=g) * (hi - lo) + lo
    b[:, 3] = b[:, 2] * (0.4 + 0.5 * torch.rand(k, generator=g))  # w >= h: 'le90' long-edge form
    b[:, 4] = (torch.rand(k, generator=g) - 0.5) * 3.0
    return b


def rpn_case(seed=0, extent=128, strides=(4, 8, 16), B=2, ks=(5, 3), A=3, num=64, pos_fraction=0.5, scale=4):
    """-> dict with everything the reference / oracle / HIP path need; anchors of size `scale * stride`"""
    g = torch.Generator().manual_seed(seed)
    sizes = [(extent // s, extent // s) for s in strides]
    anchors = cpu_grid_anchors(sizes, list(strides), scales=(scale,))
    flat = torch.cat(anchors)
    inside = LO.anchor_inside_flags(flat, torch.ones(flat.shape[0], dtype=torch.bool), (extent, extent, 3), 0)
    gts = [oriented_gts(k, extent, seed * 10 + i) for i, k in enumerate(ks)]
    cls = [torch.randn(B, A, h, w, generator=g) for h, w in sizes]
    reg = [torch.randn(B, 6 * A, h, w, generator=g) * 0.5 for h, w in sizes]
    picks, gt_inds_all = [], []
    for i in range(B):
        hb = rpn_oracle.obb2xyxy_le90(gts[i])
        gi, _, _, _ = assign_oracle.max_iou_assign(flat[inside].numpy(), hb.numpy(), False, **RPN_ASSIGN)
        full = torch.full((flat.shape[0],), -1, dtype=torch.long)
        full[inside] = torch.from_numpy(gi)
        gt_inds_all.append(full)
        pos, neg = (full > 0).nonzero().squeeze(1), (full == 0).nonzero().squeeze(1)
        npos = min(int(num * pos_fraction), pos.numel())
        pos = pos[torch.randperm(pos.numel(), generator=g)[:npos]]
        neg = neg[torch.randperm(neg.numel(), generator=g)[:num - npos]]
        picks.append((pos, neg))
    return dict(sizes=sizes, strides=strides, anchors=anchors, flat=flat, inside=inside, gts=gts, cls=cls, reg=reg,
                picks=picks, gt_inds=gt_inds_all, extent=extent, A=A, num=num, scale=scale)


def rcnn_case(seed=0, extent=256, B=2, ks=(6, 4), P=200, num=64, pos_fraction=0.25, C=26):
    """proposals around the gts (so positives exist), sampled [positives | negatives] per image, random head outputs"""
    g = torch.Generator().manual_seed(seed)
    gts = [oriented_gts(k, extent, seed * 10 + 5 + i, 16.0, 64.0) for i, k in enumerate(ks)]
    labels = [torch.randint(0, C, (k,), generator=g) for k in ks]
    pos_b, neg_b, pos_g, pos_l = [], [], [], []
    for i in range(B):
        k = ks[i]
        src = gts[i][torch.randint(0, k, (P,), generator=g)]
        jit = torch.randn(P, 5, generator=g) * torch.tensor([6.0, 6.0, 4.0, 3.0, 0.15])
        props = src + jit
        props[:, 2:4] = props[:, 2:4].clamp(min=4.0)
        props[P // 2:, :2] = torch.rand(P - P // 2, 2, generator=g) * extent  # far-away half: negatives
        boxes = torch.cat([gts[i], props])  # add_gt_as_proposals
        gi, _, lab, _ = assign_oracle.max_iou_assign(props.numpy(), gts[i].numpy(), True, gt_labels=labels[i].numpy(),
                                                     **RCNN_ASSIGN)
        gi = torch.cat([torch.arange(1, k + 1), torch.from_numpy(gi)])
        pos, neg = (gi > 0).nonzero().squeeze(1), (gi == 0).nonzero().squeeze(1)
        npos = min(int(num * pos_fraction), pos.numel())
        pos = pos[torch.randperm(pos.numel(), generator=g)[:npos]]
        neg = neg[torch.randperm(neg.numel(), generator=g)[:num - npos]]
        pos_b.append(boxes[pos]); neg_b.append(boxes[neg])
        pos_g.append(gts[i][gi[pos] - 1]); pos_l.append(labels[i][gi[pos] - 1])
    n = sum(p.shape[0] + q.shape[0] for p, q in zip(pos_b, neg_b))
    cls_score = torch.randn(n, C + 1, generator=g)
    bbox_pred = torch.randn(n, 5, generator=g) * 0.5
    return dict(gts=gts, labels=labels, pos_bboxes=pos_b, neg_bboxes=neg_b, pos_gt_bboxes=pos_g, pos_gt_labels=pos_l,
                cls_score=cls_score, bbox_pred=bbox_pred, C=C)
