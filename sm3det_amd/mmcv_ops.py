"""Host-side mirror of the reference's operator front-ends (``mmcv.ops``) for the SM3Det hot set.

Same public names, argument meaning and error behaviour as the reference wrappers, so that heads written against
``mmcv.ops`` (``getattr(ops, 'RoIAlignRotated')``: mmrotate/models/roi_heads/roi_extractors/
rotate_single_level_roi_extractor.py:59-66) work unchanged:

* ``RoIAlignRotated`` / ``roi_align_rotated``      mmcv/mmcv/ops/roi_align_rotated.py:15-177
* ``box_iou_rotated``                              mmcv/mmcv/ops/box_iou_rotated.py:9-148
* ``nms`` / ``batched_nms`` / ``nms_rotated``      mmcv/mmcv/ops/nms.py:125-183,264-382,422-477

All arithmetic happens in the gfx950 kernels behind ``sm3det_amd.mmcv_ext`` (= the ``mmcv._ext`` surface).
"""
import warnings

import numpy as np
import torch
import torch.nn as nn
from torch.autograd import Function
from torch.nn.modules.utils import _pair

from . import mmcv_ext as ext_module


# ------------------------------------------------------------------------------------------------ RoIAlignRotated
class RoIAlignRotatedFunction(Function):
    """forward/backward contract of mmcv/mmcv/ops/roi_align_rotated.py:43-104."""

    @staticmethod
    def forward(ctx, input, rois, output_size, spatial_scale, sampling_ratio=0, aligned=True, clockwise=False):
        ctx.output_size = _pair(output_size)
        ctx.spatial_scale = spatial_scale
        ctx.sampling_ratio = sampling_ratio
        ctx.aligned = aligned
        ctx.clockwise = clockwise
        ctx.save_for_backward(rois)
        ctx.feature_size = input.size()
        ctx.channels_last = (not input.is_contiguous()) and input.is_contiguous(memory_format=torch.channels_last)
        if not (input.is_contiguous() or ctx.channels_last):
            input = input.contiguous()
        output = input.new_zeros(rois.size(0), input.size(1), ctx.output_size[0], ctx.output_size[1])
        ext_module.roi_align_rotated_forward(
            input, rois.contiguous(), output, pooled_height=ctx.output_size[0], pooled_width=ctx.output_size[1],
            spatial_scale=ctx.spatial_scale, sampling_ratio=ctx.sampling_ratio, aligned=ctx.aligned,
            clockwise=ctx.clockwise)
        return output

    @staticmethod
    def backward(ctx, grad_output):
        rois = ctx.saved_tensors[0]
        batch_size, num_channels, data_height, data_width = ctx.feature_size
        out_h, out_w = grad_output.size(2), grad_output.size(3)
        grad_input = None
        if ctx.needs_input_grad[0]:
            grad_input = rois.new_zeros(batch_size, num_channels, data_height, data_width)
            if ctx.channels_last:
                grad_input = grad_input.contiguous(memory_format=torch.channels_last)
            ext_module.roi_align_rotated_backward(
                grad_output.contiguous(), rois.contiguous(), grad_input, pooled_height=out_h, pooled_width=out_w,
                spatial_scale=ctx.spatial_scale, sampling_ratio=ctx.sampling_ratio, aligned=ctx.aligned,
                clockwise=ctx.clockwise)
        return grad_input, None, None, None, None, None, None


roi_align_rotated = RoIAlignRotatedFunction.apply

_DEPRECATED_ROI_KW = {'out_size': 'output_size', 'sample_num': 'sampling_ratio'}


class RoIAlignRotated(nn.Module):
    """RoI align pooling for rotated proposals ``(batch_index, cx, cy, w, h, angle_rad)``.

    Accepts the deprecated keyword aliases ``out_size`` / ``sample_num`` the SM3Det configs use
    (local_configs/main_SM3Det.py:73-76; mmcv/mmcv/ops/roi_align_rotated.py:153-158)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for old, new in _DEPRECATED_ROI_KW.items():
            if old in kwargs:
                if new in kwargs:
                    raise AssertionError(f'The expected behavior is to replace the deprecated key `{old}` to new '
                                         f'key `{new}`, but got them in the arguments at the same time.')
                warnings.warn(f'"{old}" is deprecated in `RoIAlignRotated`, please use "{new}" instead',
                              DeprecationWarning)
                kwargs[new] = kwargs.pop(old)
        names = ['output_size', 'spatial_scale', 'sampling_ratio', 'aligned', 'clockwise']
        vals = dict(sampling_ratio=0, aligned=True, clockwise=False)
        vals.update(dict(zip(names, args)))
        for k, v in kwargs.items():
            if k not in names:
                raise TypeError(f"__init__() got an unexpected keyword argument '{k}'")
            vals[k] = v
        self.output_size = _pair(vals['output_size'])
        self.spatial_scale = float(vals['spatial_scale'])
        self.sampling_ratio = int(vals['sampling_ratio'])
        self.aligned = vals['aligned']
        self.clockwise = vals['clockwise']

    def forward(self, input, rois):
        return RoIAlignRotatedFunction.apply(input, rois, self.output_size, self.spatial_scale,
                                             self.sampling_ratio, self.aligned, self.clockwise)

    def __repr__(self):
        return (f'{self.__class__.__name__}(output_size={self.output_size}, spatial_scale={self.spatial_scale}, '
                f'sampling_ratio={self.sampling_ratio}, aligned={self.aligned}, clockwise={self.clockwise})')


# ------------------------------------------------------------------------------------------------ box_iou_rotated
def box_iou_rotated(bboxes1, bboxes2, mode='iou', aligned=False, clockwise=True):
    """IoU / IoF of rotated boxes ``(cx,cy,w,h,theta)``; (N,M) matrix or (N,) when ``aligned``."""
    assert mode in ['iou', 'iof']
    mode_flag = {'iou': 0, 'iof': 1}[mode]
    rows, cols = bboxes1.size(0), bboxes2.size(0)
    ious = bboxes1.new_zeros(rows) if aligned else bboxes1.new_zeros(rows * cols)
    if not clockwise:
        flip_mat = bboxes1.new_ones(bboxes1.shape[-1])
        flip_mat[-1] = -1
        bboxes1 = bboxes1 * flip_mat
        bboxes2 = bboxes2 * flip_mat
    ext_module.box_iou_rotated(bboxes1.contiguous(), bboxes2.contiguous(), ious, mode_flag=mode_flag,
                               aligned=aligned)
    if not aligned:
        ious = ious.view(rows, cols)
    return ious


# ------------------------------------------------------------------------------------------------ nms family
class NMSop(Function):
    """mmcv/mmcv/ops/nms.py:16-34"""

    @staticmethod
    def forward(ctx, bboxes, scores, iou_threshold, offset, score_threshold, max_num):
        is_filtering_by_score = score_threshold > 0
        if is_filtering_by_score:
            valid_mask = scores > score_threshold
            bboxes, scores = bboxes[valid_mask], scores[valid_mask]
            valid_inds = torch.nonzero(valid_mask, as_tuple=False).squeeze(dim=1)
        inds = ext_module.nms(bboxes.contiguous(), scores.contiguous(), iou_threshold=float(iou_threshold),
                              offset=offset)
        if max_num > 0:
            inds = inds[:max_num]
        if is_filtering_by_score:
            inds = valid_inds[inds]
        return inds


def nms(boxes, scores, iou_threshold=None, offset=0, score_threshold=0, max_num=-1, iou_thr=None):
    """Horizontal NMS; returns ``(dets (K,5), inds (K,))``.  ``iou_thr`` is the deprecated alias."""
    if iou_thr is not None:
        warnings.warn('"iou_thr" is deprecated in `nms`, please use "iou_threshold" instead', DeprecationWarning)
        assert iou_threshold is None
        iou_threshold = iou_thr
    assert isinstance(boxes, (torch.Tensor, np.ndarray))
    assert isinstance(scores, (torch.Tensor, np.ndarray))
    is_numpy = False
    if isinstance(boxes, np.ndarray):
        is_numpy = True
        boxes = torch.from_numpy(boxes).cuda()
    if isinstance(scores, np.ndarray):
        scores = torch.from_numpy(scores).cuda()
    assert boxes.size(1) == 4
    assert boxes.size(0) == scores.size(0)
    assert offset in (0, 1)
    inds = NMSop.apply(boxes, scores, iou_threshold, offset, score_threshold, max_num)
    dets = torch.cat((boxes[inds], scores[inds].reshape(-1, 1)), dim=1)
    if is_numpy:
        dets = dets.cpu().numpy()
        inds = inds.cpu().numpy()
    return dets, inds


def nms_rotated(dets, scores, iou_threshold, labels=None, clockwise=True):
    """Rotated NMS; returns ``(dets (K,6), keep_inds)``; mmcv/mmcv/ops/nms.py:422-477."""
    if dets.shape[0] == 0:
        return dets, None
    if not clockwise:
        flip_mat = dets.new_ones(dets.shape[-1])
        flip_mat[-1] = -1
        dets_cw = dets * flip_mat
    else:
        dets_cw = dets
    multi_label = labels is not None
    dets_wl = torch.cat((dets_cw, labels.unsqueeze(1)), 1) if multi_label else dets_cw
    _, order = scores.sort(0, descending=True)
    dets_sorted = dets_wl.index_select(0, order)
    keep_inds = ext_module.nms_rotated(dets_wl.contiguous(), scores.contiguous(), order, dets_sorted,
                                       iou_threshold, multi_label)
    dets = torch.cat((dets[keep_inds], scores[keep_inds].reshape(-1, 1)), dim=1)
    return dets, keep_inds


_NMS_OPS = {'nms': nms, 'nms_rotated': nms_rotated}


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    """Per-class NMS through the coordinate-offset trick; mmcv/mmcv/ops/nms.py:264-382."""
    if nms_cfg is None:
        scores, inds = scores.sort(descending=True)
        boxes = boxes[inds]
        return torch.cat([boxes, scores[:, None]], -1), inds
    nms_cfg_ = nms_cfg.copy()
    class_agnostic = nms_cfg_.pop('class_agnostic', class_agnostic)
    if class_agnostic:
        boxes_for_nms = boxes
    elif boxes.size(-1) == 5:
        max_coordinate = boxes[..., :2].max() + boxes[..., 2:4].max()
        offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
        boxes_ctr_for_nms = boxes[..., :2] + offsets[:, None]
        boxes_for_nms = torch.cat([boxes_ctr_for_nms, boxes[..., 2:5]], dim=-1)
    else:
        max_coordinate = boxes.max()
        offsets = idxs.to(boxes) * (max_coordinate + torch.tensor(1).to(boxes))
        boxes_for_nms = boxes + offsets[:, None]
    nms_type = nms_cfg_.pop('type', 'nms')
    if nms_type not in _NMS_OPS:
        raise NotImplementedError(f'nms type {nms_type!r} is outside the SM3Det hot path')
    nms_op = _NMS_OPS[nms_type]
    split_thr = nms_cfg_.pop('split_thr', 10000)
    if boxes_for_nms.shape[0] < split_thr:
        dets, keep = nms_op(boxes_for_nms, scores, **nms_cfg_)
        boxes = boxes[keep]
        scores = dets[:, -1]
    else:
        max_num = nms_cfg_.pop('max_num', -1)
        total_mask = scores.new_zeros(scores.size(), dtype=torch.bool)
        scores_after_nms = scores.new_zeros(scores.size())
        for id in torch.unique(idxs):
            mask = (idxs == id).nonzero(as_tuple=False).view(-1)
            dets, keep = nms_op(boxes_for_nms[mask], scores[mask], **nms_cfg_)
            total_mask[mask[keep]] = True
            scores_after_nms[mask[keep]] = dets[:, -1]
        keep = total_mask.nonzero(as_tuple=False).view(-1)
        scores, inds = scores_after_nms[keep].sort(descending=True)
        keep = keep[inds]
        boxes = boxes[keep]
        if max_num > 0:
            keep = keep[:max_num]
            boxes = boxes[:max_num]
            scores = scores[:max_num]
    boxes = torch.cat([boxes, scores[:, None]], -1)
    return boxes, keep
