"""CPU oracle of the RoI path -- TEST INFRASTRUCTURE ONLY.

* ``extract``            -- RotatedSingleRoIExtractor.forward, /root/reference/mmrotate/models/roi_heads/roi_extractors/
  rotate_single_level_roi_extractor.py:103-140, with ``map_roi_levels`` (:66-84) in torch and the per-level
  RoIAlignRotated of the plain-C oracle (oracle/ops_oracle.c, itself bit-exact vs the compiled reference op);
* ``extract_backward``   -- its gradient w.r.t. every level (the reference gets it from autograd: index_put backward ->
  RoIAlignRotated backward per level);
* ``shared2fc_forward``  -- RotatedConvFCBBoxHead.forward for the Shared2FC form, convfc_rbbox_head.py:162-201.
Pinned: tests/test_oracle_heads_live.py runs the reference's own ``RotatedSingleRoIExtractor`` (over a stand-in for its
mmdet base class's constructor bookkeeping, with the reference's own ``RoIAlignRotated`` wrapper and the reference's CPU
C++ compiled by oracle/build_ref.py underneath) and ``RotatedShared2FCBBoxHead`` live: forward bit-identical, the
extractor's backward (through the reference's autograd Function) within 1e-5."""
import numpy as np
import torch
import torch.nn.functional as F

from oracle import ops_oracle as OO


def map_roi_levels(rois, num_levels, finest_scale=56):
    scale = torch.sqrt(rois[:, 3] * rois[:, 4])
    target_lvls = torch.floor(torch.log2(scale / finest_scale + 1e-6))
    return target_lvls.clamp(min=0, max=num_levels - 1).long()


def extract(feats, rois, strides, out_size=7, sampling_ratio=2, aligned=True, clockwise=True, finest_scale=56):
    """feats: list of NCHW tensors; rois (k,6) -> (k, C, out, out), levels (k)"""
    lv = map_roi_levels(rois, len(feats), finest_scale)
    out = feats[0].new_zeros(rois.size(0), feats[0].size(1), out_size, out_size)
    for i, f in enumerate(feats):
        inds = (lv == i).nonzero(as_tuple=False).squeeze(1)
        if inds.numel() > 0:
            o = OO.roi_align_rotated_forward(f.numpy(), rois[inds].numpy(), out_size, out_size, 1.0 / strides[i],
                                             sampling_ratio, aligned, clockwise)
            out[inds] = torch.from_numpy(np.asarray(o))
    return out, lv


def extract_backward(gout, feats_shapes, rois, strides, out_size=7, sampling_ratio=2, aligned=True, clockwise=True,
                     finest_scale=56):
    lv = map_roi_levels(rois, len(feats_shapes), finest_scale)
    grads = []
    for i, shp in enumerate(feats_shapes):
        inds = (lv == i).nonzero(as_tuple=False).squeeze(1)
        if inds.numel() > 0:
            g = OO.roi_align_rotated_backward(gout[inds].contiguous().numpy(), rois[inds].numpy(), tuple(shp),
                                              out_size, out_size, 1.0 / strides[i], sampling_ratio, aligned,
                                              clockwise)
            grads.append(torch.from_numpy(np.asarray(g)))
        else:
            grads.append(torch.zeros(shp))
    return grads


def shared2fc_forward(x, p):
    x = x.flatten(1)
    for i in range(2):
        x = F.relu(F.linear(x, p[f'shared_fcs.{i}.weight'], p[f'shared_fcs.{i}.bias']))
    return F.linear(x, p['fc_cls.weight'], p['fc_cls.bias']), F.linear(x, p['fc_reg.weight'], p['fc_reg.bias'])


# ---------------------------------------------------------------------------------------------------------------------
# test-time path of the two-stage branches (CPU restatement; pinned on the reference's own functions run live by
# tests/test_oracle_heads_live.py::test_roi_head_simple_test_vs_live_reference)
def multiclass_nms_rotated(multi_bboxes, multi_scores, score_thr, iou_thr, max_num=-1):
    """mmrotate/core/post_processing/bbox_nms_rotated.py:6-96 in numpy over the plain-C ``nms_rotated`` oracle: drop the
    background column, keep scores > score_thr, shift every class by label * (max(x, y) + max(w, h) + 1), one NMS, the first
    max_num.  -> (dets (k, 6) float32, labels (k,) int64, kept indices into the filtered candidate list)"""
    mb, ms = np.asarray(multi_bboxes, np.float32), np.asarray(multi_scores, np.float32)
    n, nc = ms.shape[0], ms.shape[1] - 1
    bboxes = mb.reshape(n, -1, 5) if mb.shape[1] > 5 else np.broadcast_to(mb[:, None], (n, nc, 5))
    scores = ms[:, :-1].reshape(-1)
    labels = np.broadcast_to(np.arange(nc, dtype=np.int64)[None], (n, nc)).reshape(-1)
    bboxes = bboxes.reshape(-1, 5)
    inds = np.nonzero(scores > np.float32(score_thr))[0]
    bboxes, scores, labels = np.ascontiguousarray(bboxes[inds]), np.ascontiguousarray(scores[inds]), labels[inds]
    if bboxes.size == 0:
        return np.zeros((0, 6), np.float32), labels, inds
    max_coordinate = np.float32(bboxes[:, :2].max() + bboxes[:, 2:4].max())
    offsets = labels.astype(np.float32) * np.float32(max_coordinate + np.float32(1))
    shifted = bboxes.copy()
    shifted[:, :2] = shifted[:, :2] + offsets[:, None]
    keep = np.asarray(OO.nms_rotated(shifted, scores, float(iou_thr)), dtype=np.int64)
    if max_num > 0:
        keep = keep[:max_num]
    return np.concatenate([bboxes[keep], scores[keep][:, None]], 1), labels[keep], keep


def get_bboxes(rois, cls_score, bbox_pred, img_shape, scale_factor, rescale, cfg, means, stds, coder_kw):
    """RotatedBBoxHead.get_bboxes, rotated_bbox_head.py:358-430 (softmax, DeltaXYWHAOBBoxCoder.decode clipped to
    img_shape, optional rescale, multiclass rotated NMS)"""
    from oracle import rpn_oracle as RPO
    scores = F.softmax(cls_score, dim=-1)
    bboxes = RPO.xywha_delta2bbox(rois[:, 1:], bbox_pred, means, stds, max_shape=img_shape, **coder_kw)
    if rescale and bboxes.shape[0] > 0:
        sf = bboxes.new_tensor(scale_factor)
        bboxes = bboxes.view(bboxes.size(0), -1, 5).clone()
        bboxes[..., :4] = bboxes[..., :4] / sf
        bboxes = bboxes.view(bboxes.size(0), -1)
    if cfg is None:
        return bboxes, scores
    return multiclass_nms_rotated(bboxes.numpy(), scores.numpy(), cfg['score_thr'], cfg['nms']['iou_thr'], cfg['max_per_img'])[:2]


def simple_test(feats, proposals, img_metas, head_params, strides, num_classes, cfg, means, stds, coder_kw, rescale=False):
    """OrientedStandardRoIHead.simple_test (rotate_standard_roi_head.py:235-262 over oriented_standard_roi_head.py:126-188)
    -> per image a list over the classes of (k, 6) float32 arrays"""
    rois = torch.cat([torch.cat([torch.full((p.shape[0], 1), float(i)), p[:, :5]], 1) for i, p in enumerate(proposals)])
    x, _ = extract(feats, rois, strides)
    cs, bp = shared2fc_forward(x, head_params)
    out, start = [], 0
    for i, p in enumerate(proposals):
        sl = slice(start, start + p.shape[0])
        start += p.shape[0]
        dets, labels = get_bboxes(rois[sl], cs[sl], bp[sl], img_metas[i].get('img_shape'), img_metas[i].get('scale_factor'),
                                  rescale, cfg, means, stds, coder_kw)
        dets, labels = np.asarray(dets, np.float32), np.asarray(labels)
        out.append([dets[labels == c] for c in range(num_classes)] if dets.shape[0] else
                   [np.zeros((0, 6), np.float32) for _ in range(num_classes)])
    return out
