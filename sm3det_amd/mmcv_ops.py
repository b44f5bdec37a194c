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
# Same call surface as mmcv/mmcv/ops/nms.py (``nms`` :125-183, ``nms_rotated`` :422-477, ``batched_nms`` :264-382) -- the
# names, argument meaning, return pairs and the native calls issued are the reference's (tests/test_ref_wrappers.py checks
# the native calls one for one against the reference's own wrappers) -- but the bodies are this package's: no autograd
# Function around an index-returning operator, and ONE kernel launch per ``batched_nms`` whatever the box count.
def _as_device_tensor(x):
    """the reference accepts numpy arrays and moves them to the GPU (nms.py:160-166)"""
    if isinstance(x, np.ndarray):
        return torch.from_numpy(x).cuda(), True
    if not isinstance(x, torch.Tensor):
        raise AssertionError(f'expected a torch.Tensor or numpy array, got {type(x)}')
    return x, False


def nms(boxes, scores, iou_threshold=None, offset=0, score_threshold=0, max_num=-1, iou_thr=None):
    """Horizontal NMS of ``(x1, y1, x2, y2)`` boxes -> ``(dets (K,5) = boxes | score, kept indices (K,))`` in descending
    score order.  ``iou_thr`` is the deprecated spelling of ``iou_threshold``; ``score_threshold`` drops boxes before the
    operator runs, ``max_num`` truncates the keep list."""
    if iou_thr is not None:
        if iou_threshold is not None:
            raise AssertionError('pass iou_threshold or the deprecated iou_thr, not both')
        warnings.warn('"iou_thr" is deprecated in `nms`, please use "iou_threshold" instead', DeprecationWarning)
        iou_threshold = iou_thr
    boxes, from_numpy = _as_device_tensor(boxes)
    scores, _ = _as_device_tensor(scores)
    if boxes.dim() != 2 or boxes.size(1) != 4 or boxes.size(0) != scores.size(0) or offset not in (0, 1):
        raise AssertionError(f'nms: boxes (N,4), scores (N,), offset 0|1; got {tuple(boxes.shape)}, '
                             f'{tuple(scores.shape)}, {offset}')
    survivors = None
    cand_boxes, cand_scores = boxes, scores
    if score_threshold > 0:  # pre-filter: the operator then indexes the filtered list
        above = scores > score_threshold
        survivors = above.nonzero(as_tuple=False).squeeze(1)
        cand_boxes, cand_scores = boxes[above], scores[above]
    kept = ext_module.nms(cand_boxes.contiguous(), cand_scores.contiguous(), iou_threshold=float(iou_threshold),
                          offset=offset)
    if max_num > 0:
        kept = kept[:max_num]
    if survivors is not None:
        kept = survivors[kept]
    dets = torch.cat((boxes[kept], scores[kept].unsqueeze(1)), dim=1)
    return (dets.cpu().numpy(), kept.cpu().numpy()) if from_numpy else (dets, kept)


def nms_rotated(dets, scores, iou_threshold, labels=None, clockwise=True):
    """Rotated NMS of ``(cx, cy, w, h, angle)`` boxes -> ``(dets (K,6), kept indices)``; with ``labels`` only boxes of one
    label suppress each other.  Counter-clockwise angles are negated for the operator, the returned boxes are the caller's."""
    if dets.shape[0] == 0:
        return dets, None
    op_boxes = dets
    if not clockwise:
        sign = dets.new_ones(dets.shape[-1])
        sign[-1] = -1
        op_boxes = dets * sign
    with_labels = labels is not None
    if with_labels:
        op_boxes = torch.cat((op_boxes, labels.unsqueeze(1)), 1)
    order = scores.sort(0, descending=True)[1]
    kept = ext_module.nms_rotated(op_boxes.contiguous(), scores.contiguous(), order, op_boxes.index_select(0, order),
                                  iou_threshold, with_labels)
    return torch.cat((dets[kept], scores[kept].unsqueeze(1)), dim=1), kept


_NMS_OPS = {'nms': nms, 'nms_rotated': nms_rotated}


def _class_separated(boxes, idxs):
    """boxes moved so that different classes cannot overlap: class c is shifted by c * (extent + 1) along both axes (rotated
    boxes: the centre is shifted, extent = largest centre coordinate + largest side)"""
    if boxes.size(-1) == 5:
        extent = boxes[..., :2].max() + boxes[..., 2:4].max()
        shift = idxs.to(boxes) * (extent + torch.tensor(1).to(boxes))
        return torch.cat([boxes[..., :2] + shift[:, None], boxes[..., 2:5]], dim=-1)
    shift = idxs.to(boxes) * (boxes.max() + torch.tensor(1).to(boxes))
    return boxes + shift[:, None]


def batched_nms(boxes, scores, idxs, nms_cfg, class_agnostic=False):
    """NMS within each class ``idxs`` -> ``(boxes | score (K,5|6), kept indices)``, descending score.

    The reference loops over the classes (one operator call and a host read of ``torch.unique`` per class) once the box count
    reaches ``split_thr``; both of its paths run the operator on the class-separated boxes, where boxes of different classes
    never intersect, so they keep the same set -- here it is always ONE call (``split_thr`` is accepted and ignored)."""
    if nms_cfg is None:
        scores, order = scores.sort(descending=True)
        return torch.cat([boxes[order], scores[:, None]], -1), order
    cfg = dict(nms_cfg)
    class_agnostic = cfg.pop('class_agnostic', class_agnostic)
    kind = cfg.pop('type', 'nms')
    if kind not in _NMS_OPS:
        raise NotImplementedError(f'nms type {kind!r} is outside the SM3Det hot path')
    cfg.pop('split_thr', None)
    max_num = cfg.pop('max_num', -1)
    if kind == 'nms' and max_num > 0:
        cfg['max_num'] = max_num  # the horizontal operator truncates itself (nms.py:150)
    dets, kept = _NMS_OPS[kind](boxes if class_agnostic else _class_separated(boxes, idxs), scores, **cfg)
    if kind != 'nms' and max_num > 0 and kept is not None:
        dets, kept = dets[:max_num], kept[:max_num]
    if kept is None:  # nms_rotated on an empty input
        return torch.cat([boxes, scores[:, None]], -1), kept
    return torch.cat([boxes[kept], dets[:, -1:]], -1), kept
