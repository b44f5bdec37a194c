"""oracle/assign_oracle.py -- numpy restatement of mmdet 2.x ``MaxIoUAssigner.assign`` /
``assign_wrt_overlaps`` (core/bbox/assigners/max_iou_assigner.py) with ``BboxOverlaps2D`` (bbox_overlaps, mode 'iou',
eps 1e-6) or ``RBboxOverlaps2D`` (mmrotate rotate_iou2d_calculator.py:52-87: w,h clamped to >= 1e-3, box_iou_rotated).

TEST INFRASTRUCTURE ONLY.  The assigner rule (negatives `0 <= max < neg_thr`, positives `max >= pos_thr` -> argmax+1, then
for each gt in order `overlaps[i] == gt_max[i] >= min_pos_iou` -> i+1) is pinned on the copy of it the reference tree
carries -- ``MaxConvexIoUAssigner.assign_wrt_overlaps``, mmrotate/core/bbox/assigners/max_convex_iou_assigner.py:124-207,
run live by tests/test_oracle_heads_live.py (the ``MaxIoUAssigner`` class the configs name is mmdet's and is not vendored;
its horizontal ``bbox_overlaps`` is restated here from the published formula and is NOT pinned).  The rotated IoU
underneath is pinned: oracle/ops_oracle.c, bit-exact against the compiled reference."""
import numpy as np

from oracle import ops_oracle


def bbox_overlaps(g, b):
    """(k,4) x (n,4) -> (k,n) float32, mmdet bbox_overlaps mode='iou'"""
    g, b = g.astype(np.float32), b.astype(np.float32)
    a1 = (g[:, 2] - g[:, 0]) * (g[:, 3] - g[:, 1])
    a2 = (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])
    lt = np.maximum(g[:, None, :2], b[None, :, :2])
    rb = np.minimum(g[:, None, 2:4], b[None, :, 2:4])
    wh = np.clip(rb - lt, 0, None).astype(np.float32)
    ov = wh[..., 0] * wh[..., 1]
    uni = np.maximum(a1[:, None] + a2[None, :] - ov, np.float32(1e-6))
    return (ov / uni).astype(np.float32)


def rbbox_overlaps(g, b):
    g, b = g[:, :5].astype(np.float32).copy(), b[:, :5].astype(np.float32).copy()
    g[:, 2:4] = np.maximum(g[:, 2:4], np.float32(1e-3))
    b[:, 2:4] = np.maximum(b[:, 2:4], np.float32(1e-3))
    return ops_oracle.box_iou_rotated(g, b, 0)


def max_iou_assign(bboxes, gts, rotated, pos_iou_thr, neg_iou_thr, min_pos_iou, match_low_quality=True, gt_labels=None):
    n, k = len(bboxes), len(gts)
    gt_inds = np.full(n, -1, np.int64)
    if k == 0 or n == 0:
        gt_inds[:] = 0
        return gt_inds, np.zeros(n, np.float32), (np.full(n, -1, np.int64) if gt_labels is not None else None), None
    ov = rbbox_overlaps(gts, bboxes) if rotated else bbox_overlaps(gts[:, :4], bboxes[:, :4])
    max_ov, argmax = ov.max(0), ov.argmax(0)
    gt_max = ov.max(1)
    gt_inds[(max_ov >= 0) & (max_ov < neg_iou_thr)] = 0
    pos = max_ov >= pos_iou_thr
    gt_inds[pos] = argmax[pos] + 1
    if match_low_quality:
        for i in range(k):
            if gt_max[i] >= min_pos_iou:
                gt_inds[ov[i] == gt_max[i]] = i + 1
    labels = None
    if gt_labels is not None:
        labels = np.full(n, -1, np.int64)
        p = gt_inds > 0
        labels[p] = gt_labels[gt_inds[p] - 1]
    return gt_inds, max_ov.astype(np.float32), labels, ov
