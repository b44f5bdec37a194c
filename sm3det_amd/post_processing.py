"""Test-time post-processing of the two-stage (RGB / IR) branches and the result containers of all three.

Mirrors ``mmrotate/core/post_processing/bbox_nms_rotated.py:6-96`` (``multiclass_nms_rotated``: background column dropped,
score filter, the class-offset trick, ONE ``nms_rotated`` over all classes, top ``max_num``) and
``mmrotate/core/bbox/transforms.py:54-70`` (``rbbox2result``) / mmdet's ``bbox2result`` (the same rule with 5 columns).
``nms_rotated`` is this package's gfx950 kernel (``sm3_nms_rotated`` through ``mmcv_ops.nms_rotated``; keep lists
bit-exact to the reference's CPU operator, tests/test_ops_gpu.py), so for identical boxes and scores the kept set equals the
reference's index for index.  Pinned: ``tests/test_oracle_heads_live.py`` runs the reference's own function and
``RotatedBBoxHead.get_bboxes`` / ``OrientedStandardRoIHead.simple_test`` live against ``oracle/roi_oracle.py``;
``tests/test_roi_head_gpu.py`` compares this module with that oracle."""
import numpy as np
import torch

from . import mmcv_ops


def multiclass_nms_rotated(multi_bboxes, multi_scores, score_thr, nms, max_num=-1, score_factors=None, return_inds=False):
    """multi_bboxes (n, #class * 5) or (n, 5); multi_scores (n, #class + 1) with the background LAST -> (dets (k, 6),
    labels (k,)) [+ the kept indices into the flattened (n * #class) candidate list]"""
    num_classes = multi_scores.size(1) - 1
    if multi_bboxes.shape[1] > 5:
        bboxes = multi_bboxes.view(multi_scores.size(0), -1, 5)
    else:
        bboxes = multi_bboxes[:, None].expand(multi_scores.size(0), num_classes, 5)
    scores = multi_scores[:, :-1]
    labels = torch.arange(num_classes, dtype=torch.long, device=scores.device).view(1, -1).expand_as(scores)
    bboxes, scores, labels = bboxes.reshape(-1, 5), scores.reshape(-1), labels.reshape(-1)
    valid_mask = scores > score_thr  # on the raw scores, before the factors (as the reference)
    if score_factors is not None:
        scores = scores * score_factors.view(-1, 1).expand(multi_scores.size(0), num_classes).reshape(-1)
    inds = valid_mask.nonzero(as_tuple=False).squeeze(1)
    bboxes, scores, labels = bboxes[inds], scores[inds], labels[inds]
    if bboxes.numel() == 0:
        dets = torch.cat([bboxes, scores[:, None]], -1)
        return (dets, labels, inds) if return_inds else (dets, labels)
    # max(x, y) + max(w, h) bounds every polygon coordinate: boxes of different classes never overlap after the shift
    max_coordinate = bboxes[:, :2].max() + bboxes[:, 2:4].max()
    offsets = labels.to(bboxes) * (max_coordinate + 1)
    bboxes_for_nms = bboxes.clone()
    bboxes_for_nms[:, :2] = bboxes_for_nms[:, :2] + offsets[:, None]
    iou_thr = nms['iou_thr'] if isinstance(nms, dict) else nms.iou_thr
    _, keep = mmcv_ops.nms_rotated(bboxes_for_nms, scores, iou_thr)
    if max_num > 0:
        keep = keep[:max_num]
    dets = torch.cat([bboxes[keep], scores[keep][:, None]], 1)
    return (dets, labels[keep], keep) if return_inds else (dets, labels[keep])


def rbbox2result(bboxes, labels, num_classes):
    """(n, 6) detections + (n,) labels -> list over classes of (k_c, 6) float32 arrays (the reference's result type)"""
    if bboxes.shape[0] == 0:
        return [np.zeros((0, 6), dtype=np.float32) for _ in range(num_classes)]
    bboxes, labels = bboxes.detach().cpu().numpy(), labels.detach().cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes)]


def bbox2result(bboxes, labels, num_classes):
    """mmdet.core.bbox2result: the same rule for horizontal detections (n, 5)"""
    if bboxes.shape[0] == 0:
        return [np.zeros((0, bboxes.shape[1] if bboxes.dim() == 2 else 5), dtype=np.float32) for _ in range(num_classes)]
    bboxes, labels = bboxes.detach().cpu().numpy(), labels.detach().cpu().numpy()
    return [bboxes[labels == i, :] for i in range(num_classes)]
