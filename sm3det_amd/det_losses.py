"""Targets + losses of the two-stage (RGB / IR) branch on the MI355X -- SURVEY.md 8(f) row 3, the glue between target
assignment and a scalar loss.

What it mirrors (all inside the reference tree, so pinned through ``oracle/ref_heads.py``):

* ``OrientedRPNHead._get_targets_single`` + ``RotatedRPNHead.get_targets`` / ``loss`` / ``OrientedRPNHead.loss_single``
  (``mmrotate/models/dense_heads/oriented_rpn_head.py:26-187``, ``rotated_rpn_head.py:152-372``);
* ``RotatedBBoxHead._get_target_single`` / ``get_targets`` / ``loss``
  (``mmrotate/models/roi_heads/bbox_heads/rotated_bbox_head.py:141-356``).

The loss FORMULAS are mmdet 2.x's (``CrossEntropyLoss`` with / without ``use_sigmoid``, ``SmoothL1Loss``): mmdet is not
vendored by the reference, they are restated -- parity unpinned for those three formulas only.

MI355X design: the reference builds dense per-anchor target / weight tensors (261 888 x 6 per image), unmaps them,
re-splits them per level and evaluates the loss over every anchor although only the sampled <= 256 per image carry a
weight; here the sampler's fixed-size index block goes straight into ONE kernel that gathers the predictions, encodes the
targets from (anchor, matched gt) and reduces the per-level losses, and a second one that writes the gradient at the
sampled positions of zero-filled maps.  No host synchronisation anywhere (counts stay on the device).
"""
import ctypes

import torch
from torch.autograd import Function

from . import _lib
from ._lib import SM3Error, check, lib, stream_ptr

MAX_LEVELS = 8


class RpnLossLevel(ctypes.Structure):
    """mirror of `sm3_rpn_loss_level` (include/sm3det_hip.h)"""
    _fields_ = [('cls', ctypes.c_void_p), ('reg', ctypes.c_void_p),
                ('cls_stride', ctypes.c_long * 3), ('reg_stride', ctypes.c_long * 3),
                ('dcls', ctypes.c_void_p), ('dreg', ctypes.c_void_p),
                ('dcls_stride', ctypes.c_long * 3), ('dreg_stride', ctypes.c_long * 3),
                ('num_anchors', ctypes.c_long)]


class RpnLossDesc(ctypes.Structure):
    """mirror of `sm3_rpn_loss_desc`"""
    _fields_ = [('level', RpnLossLevel * MAX_LEVELS), ('num_levels', ctypes.c_int), ('anchors_per_pos', ctypes.c_int),
                ('anchors', ctypes.c_void_p), ('idx', ctypes.c_void_p), ('is_pos', ctypes.c_void_p),
                ('valid', ctypes.c_void_p), ('gt_inds', ctypes.c_void_p), ('gts', ctypes.c_void_p),
                ('batch', ctypes.c_int), ('samples', ctypes.c_int), ('total_anchors', ctypes.c_int),
                ('max_gts', ctypes.c_int), ('n_pos', ctypes.c_void_p), ('n_neg', ctypes.c_void_p),
                ('means', ctypes.c_float * 6), ('stds', ctypes.c_float * 6), ('beta', ctypes.c_float),
                ('loss_weight_cls', ctypes.c_float), ('loss_weight_bbox', ctypes.c_float), ('pos_weight', ctypes.c_float)]


class RcnnLossDesc(ctypes.Structure):
    """mirror of `sm3_rcnn_loss_desc`"""
    _fields_ = [('cls_score', ctypes.c_void_p), ('bbox_pred', ctypes.c_void_p), ('dcls_score', ctypes.c_void_p),
                ('dbbox_pred', ctypes.c_void_p), ('ld_cls', ctypes.c_int), ('ld_reg', ctypes.c_int),
                ('num_classes', ctypes.c_int), ('num_rois', ctypes.c_int), ('labels', ctypes.c_void_p),
                ('valid', ctypes.c_void_p), ('rois', ctypes.c_void_p), ('gts', ctypes.c_void_p),
                ('label_weights', ctypes.c_void_p), ('bbox_targets', ctypes.c_void_p),
                ('means', ctypes.c_float * 5), ('stds', ctypes.c_float * 5), ('norm_factor', ctypes.c_float),
                ('edge_swap', ctypes.c_int), ('proj_xy', ctypes.c_int), ('beta', ctypes.c_float),
                ('loss_weight_cls', ctypes.c_float), ('loss_weight_bbox', ctypes.c_float), ('pos_weight', ctypes.c_float)]


def signatures():
    P, I = ctypes.c_void_p, ctypes.c_int
    return {
        'sm3_obb2xyxy_le90': (I, [P, I, I, P, P]),
        'sm3_rpn_loss_forward': (I, [ctypes.POINTER(RpnLossDesc), P, P, P]),
        'sm3_rpn_loss_backward': (I, [ctypes.POINTER(RpnLossDesc), P, P, P]),
        'sm3_rcnn_loss_forward': (I, [ctypes.POINTER(RcnnLossDesc), P, P, P]),
        'sm3_rcnn_loss_backward': (I, [ctypes.POINTER(RcnnLossDesc), P, P, P, P]),
    }


def obb2xyxy(obboxes, version='le90'):
    """mmrotate.core.obb2xyxy (transforms.py:685-702) for 'le90': (n, 5+) oriented -> (n, 4) enclosing horizontal boxes"""
    if version != 'le90':
        raise NotImplementedError("only angle version 'le90' (every SM3Det config) is implemented")
    _lib.require_gpu(obboxes)
    b = obboxes.detach().float().contiguous()
    out = torch.empty(b.shape[0], 4, device=b.device)
    if b.shape[0]:
        with torch.cuda.device(b.device):
            check(lib().sm3_obb2xyxy_le90(b.data_ptr(), b.shape[0], b.shape[1], out.data_ptr(), stream_ptr()),
                  'obb2xyxy_le90')
    return out


def _strides3(t, layout):
    """(image, position, channel) element strides of a logically (B, Ch, H, W) prediction map"""
    B, Ch, H, W = t.shape
    sb, sc, sh, sw = t.stride()
    if H > 1 and sh != W * sw:
        raise SM3Error(f'{layout}: the spatial dims of a prediction map must be jointly contiguous (NHWC or NCHW)')
    return sb, sw if W > 1 or H > 1 else 1, sc


class _RPNLoss(Function):
    """(cls_0, reg_0, cls_1, reg_1, ...) -> (loss_cls (L,), loss_bbox (L,)).  Everything else is non-differentiable."""

    @staticmethod
    def forward(ctx, meta, *maps):
        (anchors, idx, is_pos, valid, gt_inds, gts, n_pos, n_neg, A, means, stds, beta, w_cls, w_bbox,
         pos_weight) = meta
        L = len(maps) // 2
        if L < 1 or L > MAX_LEVELS:
            raise SM3Error(f'RPN loss: 1..{MAX_LEVELS} levels supported, got {L}')
        cls, reg = maps[0::2], maps[1::2]
        _lib.require_gpu(anchors, idx, gt_inds, gts, *maps)
        d = RpnLossDesc()
        tot = 0
        for l in range(L):
            c, r = cls[l], reg[l]
            if c.dtype != torch.float32 or r.dtype != torch.float32:
                raise SM3Error('RPN loss: fp32 predictions expected (the reference runs this under force_fp32)')
            B, Ac, H, W = c.shape
            if Ac != A or r.shape[1] != 6 * A or r.shape[0] != B or tuple(r.shape[2:]) != (H, W):
                raise SM3Error(f'RPN loss: level {l}: cls {tuple(c.shape)} / reg {tuple(r.shape)} vs {A} anchors')
            lv = d.level[l]
            lv.cls, lv.reg = c.data_ptr(), r.data_ptr()
            lv.cls_stride[:] = _strides3(c, 'cls_score')
            lv.reg_stride[:] = _strides3(r, 'bbox_pred')
            lv.num_anchors = H * W * A
            tot += H * W * A
        B, S = idx.shape
        if tot != anchors.shape[0] or gt_inds.shape != (B, tot):
            raise SM3Error(f'RPN loss: {tot} anchors on the maps vs anchors {tuple(anchors.shape)} / gt_inds '
                           f'{tuple(gt_inds.shape)}')
        d.num_levels, d.anchors_per_pos = L, A
        d.anchors, d.idx, d.is_pos, d.valid = anchors.data_ptr(), idx.data_ptr(), is_pos.data_ptr(), valid.data_ptr()
        d.gt_inds, d.gts = gt_inds.data_ptr(), gts.data_ptr()
        d.batch, d.samples, d.total_anchors, d.max_gts = B, S, tot, gts.shape[1]
        d.n_pos, d.n_neg = n_pos.data_ptr(), n_neg.data_ptr()
        d.means[:], d.stds[:] = means, stds
        d.beta, d.loss_weight_cls, d.loss_weight_bbox, d.pos_weight = beta, w_cls, w_bbox, pos_weight
        out = torch.empty(2, L, device=anchors.device)
        with torch.cuda.device(anchors.device):
            check(lib().sm3_rpn_loss_forward(ctypes.byref(d), out[0].data_ptr(), out[1].data_ptr(), stream_ptr()),
                  'rpn_loss_forward')
        ctx.desc, ctx.keep, ctx.L, ctx.A = d, (anchors, idx, is_pos, valid, gt_inds, gts, n_pos, n_neg), L, A
        ctx.save_for_backward(*maps)
        return out[0], out[1]

    @staticmethod
    def backward(ctx, g_cls, g_bbox):
        maps = ctx.saved_tensors
        d, L, A = ctx.desc, ctx.L, ctx.A
        dev = maps[0].device
        zero = torch.zeros(L, device=dev)
        g_cls = (zero if g_cls is None else g_cls).contiguous().float()
        g_bbox = (zero if g_bbox is None else g_bbox).contiguous().float()
        grads, bufs = [], []
        for l in range(L):
            B, _, H, W = maps[2 * l].shape
            # one NHWC map per level: [d cls (A) | d reg (6A) | 0 ...] with the channel count of the map the predictions are
            # views of (the RPN tower's fused, padded head output: rpn_head._SplitClsReg then hands this buffer on as the
            # gradient of that output without a copy); 7A channels when the predictions are separate tensors
            cs, rs_ = maps[2 * l], maps[2 * l + 1]
            Cp = cs.stride(3) if (W > 1 and cs.stride(1) == 1 and rs_.stride() == cs.stride() and cs.stride(3) >= 7 * A
                                  and rs_.storage_offset() == cs.storage_offset() + A) else 7 * A
            buf = torch.zeros(B, H, W, Cp, device=dev)
            buf._sm3_rest_is_zero = True
            bufs.append(buf)
            lv = d.level[l]
            lv.dcls, lv.dreg = buf.data_ptr(), buf.data_ptr() + 4 * A
            lv.dcls_stride[:] = (H * W * Cp, Cp, 1)
            lv.dreg_stride[:] = (H * W * Cp, Cp, 1)
            grads += [buf[..., :A].permute(0, 3, 1, 2), buf[..., A:7 * A].permute(0, 3, 1, 2)]
        with torch.cuda.device(dev):
            check(lib().sm3_rpn_loss_backward(ctypes.byref(d), g_cls.data_ptr(), g_bbox.data_ptr(), stream_ptr()),
                  'rpn_loss_backward')
        return (None,) + tuple(grads)


def rpn_loss(cls_scores, bbox_preds, anchors, idx, is_pos, valid, gt_inds, gts, n_pos, n_neg, num_anchors,
             means, stds, beta=1.0 / 9.0, loss_weight_cls=1.0, loss_weight_bbox=1.0, pos_weight=-1.0):
    """Per-level (loss_rpn_cls (L,), loss_rpn_bbox (L,)) of the sampled anchors.  cls_scores[l] (B, A, H, W),
    bbox_preds[l] (B, 6A, H, W) -- any NHWC / NCHW strides; anchors (sum_l H W A, 4); idx / is_pos / valid (B, S) from
    ``RandomSampler.sample_fixed``; gt_inds (B, total) from the assigner; gts (B, Kmax, 5) oriented; n_pos / n_neg (B,)."""
    meta = (anchors.float().contiguous(), idx.long().contiguous(), is_pos.to(torch.uint8).contiguous(),
            valid.to(torch.uint8).contiguous(), gt_inds.long().contiguous(), gts.float().contiguous(),
            n_pos.long().contiguous(), n_neg.long().contiguous(), int(num_anchors),
            tuple(float(v) for v in means), tuple(float(v) for v in stds), float(beta), float(loss_weight_cls),
            float(loss_weight_bbox), float(pos_weight))
    maps = []
    for c, r in zip(cls_scores, bbox_preds):
        maps += [c, r]
    return _RPNLoss.apply(meta, *maps)


class _RCNNLoss(Function):
    @staticmethod
    def forward(ctx, cls_score, bbox_pred, meta):
        labels, valid, rois, gts, C, coder, beta, w_cls, w_bbox, pos_weight, label_w, targets = meta
        _lib.require_gpu(cls_score, bbox_pred, labels)
        if cls_score.dtype != torch.float32 or bbox_pred.dtype != torch.float32:
            raise SM3Error('RCNN loss: fp32 predictions expected (the reference runs this under force_fp32)')
        N = cls_score.shape[0]
        if cls_score.shape[1] != C + 1 or bbox_pred.shape != (N, 5) or cls_score.stride(1) != 1 or bbox_pred.stride(1) != 1:
            raise SM3Error(f'RCNN loss: cls_score {tuple(cls_score.shape)} / bbox_pred {tuple(bbox_pred.shape)} '
                           f'(class-agnostic regression, {C} classes) with unit inner stride expected')
        d = RcnnLossDesc()
        d.cls_score, d.bbox_pred = cls_score.data_ptr(), bbox_pred.data_ptr()
        d.ld_cls, d.ld_reg, d.num_classes, d.num_rois = cls_score.stride(0), bbox_pred.stride(0), C, N
        p = lambda t: None if t is None else t.data_ptr()  # noqa: E731
        d.labels, d.valid, d.rois, d.gts = p(labels), p(valid), p(rois), p(gts)
        d.label_weights, d.bbox_targets = p(label_w), p(targets)
        d.means[:], d.stds[:] = coder.means, coder.stds
        d.norm_factor = float(coder.norm_factor or 0.0)
        d.edge_swap, d.proj_xy = int(coder.edge_swap), int(coder.proj_xy)
        d.beta, d.loss_weight_cls, d.loss_weight_bbox, d.pos_weight = beta, w_cls, w_bbox, pos_weight
        out = torch.empty(3, device=cls_score.device)
        counts = torch.empty(2, device=cls_score.device)
        with torch.cuda.device(cls_score.device):
            check(lib().sm3_rcnn_loss_forward(ctypes.byref(d), out.data_ptr(), counts.data_ptr(), stream_ptr()),
                  'rcnn_loss_forward')
        ctx.desc, ctx.keep = d, (labels, valid, rois, gts, label_w, targets)
        ctx.save_for_backward(cls_score, bbox_pred, counts)
        # three independent scalars (clones, not views of `out`: an in-place op on one returned loss must not alias the
        # others), and the SAME `acc` object is marked and returned -- marking a temporary view had no effect
        l_cls, l_box, acc = out[0].clone(), out[1].clone(), out[2].clone()
        ctx.mark_non_differentiable(acc)
        return l_cls, l_box, acc

    @staticmethod
    def backward(ctx, g_cls, g_bbox, _g_acc):
        cls_score, bbox_pred, counts = ctx.saved_tensors
        d = ctx.desc
        dev = cls_score.device
        zero = torch.zeros(1, device=dev)
        g_cls = (zero if g_cls is None else g_cls.reshape(1)).contiguous().float()
        g_bbox = (zero if g_bbox is None else g_bbox.reshape(1)).contiguous().float()
        dc = torch.empty(cls_score.shape[0], d.ld_cls, device=dev)
        dr = torch.empty(bbox_pred.shape[0], d.ld_reg, device=dev)
        if d.ld_cls != cls_score.shape[1]:
            dc.zero_()
        if d.ld_reg != 5:
            dr.zero_()
        d.dcls_score, d.dbbox_pred = dc.data_ptr(), dr.data_ptr()
        with torch.cuda.device(dev):
            check(lib().sm3_rcnn_loss_backward(ctypes.byref(d), counts.data_ptr(), g_cls.data_ptr(), g_bbox.data_ptr(),
                                               stream_ptr()), 'rcnn_loss_backward')
        return dc[:, :cls_score.shape[1]], dr[:, :5], None


def rcnn_loss(cls_score, bbox_pred, labels, valid, rois, gts, num_classes, coder, beta=1.0, loss_weight_cls=1.0,
              loss_weight_bbox=1.0, pos_weight=-1.0, label_weights=None, bbox_targets=None):
    """(loss_cls, loss_bbox, acc) of ``RotatedBBoxHead.loss`` with the targets of ``get_targets`` built in place.
    cls_score (N, C+1), bbox_pred (N, 5); labels (N,) in [0, C] (C = background); valid (N,); rois (N, 5) the sampled
    boxes; gts (N, 5) the matched ground truth of the positives (any finite value elsewhere).  Alternatively the
    reference API's precomputed ``label_weights`` (N,) / ``bbox_targets`` (N, 5) replace valid / (rois, gts)."""
    c = lambda t, f: None if t is None else f(t).contiguous()  # noqa: E731
    meta = (labels.long().contiguous(), c(valid, lambda t: t.to(torch.uint8)), c(rois, lambda t: t.detach().float()),
            c(gts, lambda t: t.detach().float()), int(num_classes), coder, float(beta), float(loss_weight_cls),
            float(loss_weight_bbox), float(pos_weight), c(label_weights, lambda t: t.detach().float()),
            c(bbox_targets, lambda t: t.detach().float()))
    return _RCNNLoss.apply(cls_score, bbox_pred, meta)
