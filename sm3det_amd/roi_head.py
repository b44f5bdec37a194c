"""RoI path of the 2-stage (RGB / IR) branch on the MI355X kernels: the feature
extractor and the box head's fully-connected stack.

* ``RotatedSingleRoIExtractor`` -- mirror of ``mmrotate/models/roi_heads/roi_extractors/
  rotate_single_level_roi_extractor.py`` (ctor :30-40, ``map_roi_levels`` :66-84, ``forward`` :87-140).  The reference
  loops over levels with ``nonzero()`` + gather + ``RoIAlignRotated`` + scatter (four host syncs and ~16 launches);
  here ONE launch serves every RoI at its own level (``sm3_roi_align_rotated_multilevel_*``), forward and backward,
  on the NHWC features the neck emits.
* ``RotatedShared2FCBBoxHead`` -- the layers and ``forward`` of ``mmrotate/models/roi_heads/bbox_heads/
  convfc_rbbox_head.py`` (:127-160 ``_add_conv_fc_branch``, :162-201 ``forward``, :204-218) for the Shared2FC form:
  ``shared_fcs.{0,1}`` (+ReLU, fused into the GEMM epilogue), ``fc_cls`` and ``fc_reg`` evaluated as ONE GEMM.  Same
  parameter names / shapes.  Losses, target computation and box decoding belong to mmdet / the training loop and are
  not mirrored.
No CPU fallback.
"""
import ctypes
import os

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib
from . import backbone_ops as ops
from .registry import ROTATED_NECKS as _REG


def _layout_of(t):
    if t.is_contiguous():
        return 0
    if t.is_contiguous(memory_format=torch.channels_last):
        return 1
    return -1


class _MultiLevelRoIAlign(Function):
    @staticmethod
    def forward(ctx, rois, cfg, *feats):
        out_h, out_w, sampling_ratio, aligned, clockwise, strides, finest = cfg
        _lib.require_gpu(rois, *feats)
        rois = rois.float().contiguous()
        layouts = {_layout_of(f) for f in feats}
        if layouts == {1}:
            layout = 1
        else:  # mixed / strided inputs: bring everything to NCHW-contiguous once
            feats = tuple(f.contiguous() for f in feats)
            layout = 0
        n, C, L = rois.shape[0], feats[0].shape[1], len(feats)
        if any(f.dtype != torch.float32 or f.shape[1] != C for f in feats):
            raise _lib.SM3Error('all levels must be float32 with the same channel count')
        out = torch.empty(n, C, out_h, out_w, device=rois.device)  # (every RoI has a level: the kernel writes all of it)
        levels = torch.empty(n, dtype=torch.int32, device=rois.device)
        ctx.geom = ((ctypes.c_int * L)(*[f.shape[2] for f in feats]), (ctypes.c_int * L)(*[f.shape[3] for f in feats]),
                    (ctypes.c_float * L)(*[1.0 / s for s in strides]))
        ctx.cfg, ctx.layout = cfg, layout
        ctx.shapes = [tuple(f.shape) for f in feats]
        if n:
            ptrs = (ctypes.c_void_p * L)(*[f.data_ptr() for f in feats])
            hs, ws, sc = ctx.geom
            _lib.check(_lib.lib().sm3_roi_align_rotated_multilevel_forward(
                ptrs, hs, ws, sc, L, float(finest), rois.data_ptr(), out.data_ptr(), levels.data_ptr(), n, C, out_h,
                out_w, int(sampling_ratio), int(bool(aligned)), int(bool(clockwise)), layout, _lib.stream_ptr()),
                'roi_align_rotated_multilevel_forward')
        ctx.save_for_backward(rois)
        ctx.mark_non_differentiable(levels)
        return out, levels

    @staticmethod
    def backward(ctx, gout, _glv):
        (rois,) = ctx.saved_tensors
        out_h, out_w, sampling_ratio, aligned, clockwise, strides, finest = ctx.cfg
        L = len(ctx.shapes)
        fmt = torch.channels_last if ctx.layout == 1 else torch.contiguous_format
        n = rois.shape[0]
        tiled = (n and ctx.layout == 1 and int(sampling_ratio) > 0 and n <= 65535 and ctx.shapes[0][1] % 4 == 0
                 and os.environ.get('SM3_ROI_BWD', 'tiled') == 'tiled')
        # (the gather form writes every pixel of the maps itself: no fill pass over 0.18 GB of pyramid gradients)
        grads = [(torch.empty if tiled else torch.zeros)(s, device=gout.device).contiguous(memory_format=fmt)
                 for s in ctx.shapes]
        if tiled:
            # counting sort by pixel + one gather pass per 8 x 8-pixel tile, no atomic accumulation (ops_rotated.hip, round 4)
            ptrs = (ctypes.c_void_p * L)(*[g.data_ptr() for g in grads])
            hs, ws, sc = ctx.geom
            B, C = ctx.shapes[0][0], ctx.shapes[0][1]
            lib = _lib.lib()
            nb = lib.sm3_roi_align_rotated_backward_tiled_workspace_bytes(n, B, C, out_h, out_w, int(sampling_ratio), hs, ws, L)
            wsp = _lib.workspace(nb, gout.device)
            _lib.check(lib.sm3_roi_align_rotated_backward_tiled(
                gout.contiguous().data_ptr(), rois.data_ptr(), ptrs, hs, ws, sc, L, float(finest), n, B, C, out_h, out_w,
                int(sampling_ratio), int(bool(aligned)), int(bool(clockwise)), 1, wsp.data_ptr(), nb, _lib.stream_ptr()),
                'roi_align_rotated_backward_tiled')
        elif n:
            ptrs = (ctypes.c_void_p * L)(*[g.data_ptr() for g in grads])
            hs, ws, sc = ctx.geom
            _lib.check(_lib.lib().sm3_roi_align_rotated_multilevel_backward(
                gout.contiguous().data_ptr(), rois.data_ptr(), ptrs, hs, ws, sc, L, float(finest), n,
                ctx.shapes[0][1], out_h, out_w, int(sampling_ratio), int(bool(aligned)), int(bool(clockwise)),
                ctx.layout, _lib.stream_ptr()), 'roi_align_rotated_multilevel_backward')
        return (None, None) + tuple(grads)


@_REG.register_module()
class RotatedSingleRoIExtractor(nn.Module):
    def __init__(self, roi_layer, out_channels, featmap_strides, finest_scale=56, init_cfg=None):
        super().__init__()
        cfg = dict(roi_layer)
        if cfg.pop('type', 'RoIAlignRotated') != 'RoIAlignRotated':
            raise NotImplementedError('only roi_layer type RoIAlignRotated (every SM3Det config) is implemented')
        for old, new in (('out_size', 'output_size'), ('sample_num', 'sampling_ratio')):  # deprecated mmcv aliases
            if old in cfg:
                cfg[new] = cfg.pop(old)
        size = cfg.get('output_size', 7)
        self.output_size = (size, size) if isinstance(size, int) else tuple(size)
        self.sampling_ratio = int(cfg.get('sampling_ratio', 0))
        self.aligned = bool(cfg.get('aligned', True))
        self.clockwise = bool(cfg.get('clockwise', False))
        self.out_channels = out_channels
        self.featmap_strides = list(featmap_strides)
        self.finest_scale = finest_scale
        self.fp16_enabled = False

    @property
    def num_inputs(self):
        return len(self.featmap_strides)

    def map_roi_levels(self, rois, num_levels):
        """reference :66-84 (torch ops; the fused kernel evaluates the same expression per RoI)"""
        scale = torch.sqrt(rois[:, 3] * rois[:, 4])
        lv = torch.floor(torch.log2(scale / self.finest_scale + 1e-6))
        return lv.clamp(min=0, max=num_levels - 1).long()

    def forward(self, feats, rois, roi_scale_factor=None, return_levels=False):
        if roi_scale_factor is not None:
            raise NotImplementedError('roi_scale_factor is not used by any SM3Det config')
        feats = tuple(feats[:self.num_inputs])
        cfg = (self.output_size[0], self.output_size[1], self.sampling_ratio, self.aligned, self.clockwise,
               tuple(self.featmap_strides[:len(feats)]), float(self.finest_scale))
        out, levels = _MultiLevelRoIAlign.apply(rois, cfg, *feats)
        return (out, levels) if return_levels else out


@_REG.register_module()
class RotatedShared2FCBBoxHead(nn.Module):
    def __init__(self, in_channels=256, fc_out_channels=1024, roi_feat_size=7, num_classes=80,
                 reg_class_agnostic=False, with_avg_pool=False, bbox_coder=None, loss_cls=None, loss_bbox=None,
                 init_cfg=None, **kwargs):
        super().__init__()
        if with_avg_pool:
            raise NotImplementedError('with_avg_pool=True is not used by any SM3Det config')
        self.in_channels, self.fc_out_channels, self.num_classes = in_channels, fc_out_channels, num_classes
        self.roi_feat_size = (roi_feat_size, roi_feat_size)
        self.roi_feat_area = roi_feat_size * roi_feat_size
        self.reg_class_agnostic = reg_class_agnostic
        self.num_shared_fcs = 2
        self.shared_fcs = nn.ModuleList([nn.Linear(in_channels * self.roi_feat_area, fc_out_channels),
                                         nn.Linear(fc_out_channels, fc_out_channels)])
        self.fc_cls = nn.Linear(fc_out_channels, num_classes + 1)          # reference :96-100
        self.fc_reg = nn.Linear(fc_out_channels, 5 if reg_class_agnostic else 5 * num_classes)  # :101-107
        self.bbox_coder_cfg, self.loss_cls_cfg, self.loss_bbox_cfg, self.init_cfg = (bbox_coder, loss_cls, loss_bbox,
                                                                                     init_cfg)

    def init_weights(self):
        """reference init_cfg (:108-125): Xavier/uniform on the shared fcs, Normal 0.01 / 0.001 on fc_cls / fc_reg"""
        for fc in self.shared_fcs:
            nn.init.xavier_uniform_(fc.weight)
            nn.init.constant_(fc.bias, 0)
        nn.init.normal_(self.fc_cls.weight, 0, 0.01)
        nn.init.normal_(self.fc_reg.weight, 0, 0.001)
        nn.init.constant_(self.fc_cls.bias, 0)
        nn.init.constant_(self.fc_reg.bias, 0)

    def forward(self, x):
        """reference :162-201 for num_shared_fcs = 2: flatten -> relu(fc) x 2 -> (fc_cls, fc_reg)"""
        x = x.flatten(1)
        for fc in self.shared_fcs:
            x = ops.linear_relu(x, fc.weight, fc.bias)
        nc, nr = self.fc_cls.out_features, self.fc_reg.out_features
        pad = (-(nc + nr)) % 32
        w = torch.cat([self.fc_cls.weight, self.fc_reg.weight, x.new_zeros(pad, self.fc_out_channels)], 0)
        b = torch.cat([self.fc_cls.bias, self.fc_reg.bias, x.new_zeros(pad)], 0)
        o = ops.linear(x, w, b)
        return o[:, :nc], o[:, nc:nc + nr]

    # ------------------------------------------------------------------------------------------ targets + loss
    def _coder(self):
        if getattr(self, '_bbox_coder', None) is None:
            from .rpn_head import DeltaXYWHAOBBoxCoder
            cfg = dict(self.bbox_coder_cfg or dict(type='DeltaXYWHAOBBoxCoder', angle_range='le90'))
            if cfg.pop('type', 'DeltaXYWHAOBBoxCoder') != 'DeltaXYWHAOBBoxCoder':
                raise NotImplementedError('only DeltaXYWHAOBBoxCoder (every SM3Det config) is implemented')
            self._bbox_coder = DeltaXYWHAOBBoxCoder(**cfg)
        return self._bbox_coder

    bbox_coder = property(_coder)

    def _loss_cfg(self):
        lc = dict(self.loss_cls_cfg or dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0))
        lb = dict(self.loss_bbox_cfg or dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0))
        if lc.get('type') != 'CrossEntropyLoss' or lc.get('use_sigmoid', False) or lb.get('type') != 'SmoothL1Loss':
            raise NotImplementedError('the RCNN losses of the SM3Det configs are softmax CrossEntropyLoss + SmoothL1Loss')
        if not self.reg_class_agnostic:
            raise NotImplementedError('reg_class_agnostic=False is not used by any SM3Det config')
        return float(lc.get('loss_weight', 1.0)), float(lb.get('loss_weight', 1.0)), float(lb.get('beta', 1.0))

    @torch.no_grad()
    def get_bboxes(self, rois, cls_score, bbox_pred, img_shape, scale_factor, rescale=False, cfg=None):
        """rotated_bbox_head.py:358-430: softmax scores, ``DeltaXYWHAOBBoxCoder.decode`` of the deltas on the RoIs (clipped
        to ``img_shape``), optional division of (cx, cy, w, h) by ``scale_factor``, then -- with a test cfg -- multiclass
        rotated NMS -> (dets (k, 6), labels (k,)); without one the raw (boxes, scores)."""
        from .post_processing import multiclass_nms_rotated
        scores = torch.softmax(cls_score, dim=-1) if cls_score is not None else None
        if bbox_pred is not None:
            bboxes = self._coder().decode(rois[..., 1:], bbox_pred, max_shape=img_shape)
        else:
            bboxes = rois[:, 1:].clone()
            if img_shape is not None:  # (the reference clamps a temporary here, i.e. does nothing; kept as is)
                pass
        if rescale and bboxes.size(0) > 0:
            sf = bboxes.new_tensor(scale_factor)
            bboxes = bboxes.view(bboxes.size(0), -1, 5)
            bboxes[..., :4] = bboxes[..., :4] / sf
            bboxes = bboxes.view(bboxes.size(0), -1)
        if cfg is None:
            return bboxes, scores
        return multiclass_nms_rotated(bboxes, scores, _cfg_get(cfg, 'score_thr'), _cfg_get(cfg, 'nms'),
                                      _cfg_get(cfg, 'max_per_img'))

    def get_targets(self, sampling_results, gt_bboxes, gt_labels, rcnn_train_cfg, concat=True):
        """rotated_bbox_head.py:209-273 (+ _get_target_single :141-207): per image labels / label_weights / bbox_targets /
        bbox_weights of [positives | negatives]; targets through the DeltaXYWHAOBBoxCoder encode kernel."""
        pw = float((rcnn_train_cfg or {}).get('pos_weight', -1))
        out = [[], [], [], []]
        for res in sampling_results:
            npos, nneg = res.pos_bboxes.size(0), res.neg_bboxes.size(0)
            n = npos + nneg
            dev = res.pos_bboxes.device
            labels = torch.full((n,), self.num_classes, dtype=torch.long, device=dev)
            lw = torch.zeros(n, device=dev)
            bt, bw = torch.zeros(n, 5, device=dev), torch.zeros(n, 5, device=dev)
            if npos:
                labels[:npos] = res.pos_gt_labels
                lw[:npos] = 1.0 if pw <= 0 else pw
                bt[:npos] = self.bbox_coder.encode(res.pos_bboxes[:, :5], res.pos_gt_bboxes[:, :5])
                bw[:npos] = 1
            if nneg:
                lw[-nneg:] = 1.0
            for o, v in zip(out, (labels, lw, bt, bw)):
                o.append(v)
        return tuple(torch.cat(o, 0) for o in out) if concat else tuple(out)

    def loss(self, cls_score, bbox_pred, rois, labels, label_weights, bbox_targets, bbox_weights,
             reduction_override=None):
        """rotated_bbox_head.py:275-356 on precomputed targets (bbox_weights is 1 on the positives by construction):
        dict(loss_cls, acc, loss_bbox)."""
        from . import det_losses
        if reduction_override is not None:
            raise NotImplementedError('reduction_override is not used by the SM3Det training path')
        w_cls, w_bbox, beta = self._loss_cfg()
        l_cls, l_box, acc = det_losses.rcnn_loss(cls_score.float(), bbox_pred.float(), labels, None, None, None,
                                                 self.num_classes, self.bbox_coder, beta, w_cls, w_bbox,
                                                 label_weights=label_weights, bbox_targets=bbox_targets)
        return dict(loss_cls=l_cls, acc=acc, loss_bbox=l_box)

    def loss_fused(self, cls_score, bbox_pred, labels, valid, rois, gts, pos_weight=-1.0):
        """get_targets + loss in one kernel pair on a fixed-size sample block: rois (N, 5) sampled boxes, gts (N, 5) their
        matched ground truth, labels (N,) (num_classes = background), valid (N,)."""
        from . import det_losses
        w_cls, w_bbox, beta = self._loss_cfg()
        l_cls, l_box, acc = det_losses.rcnn_loss(cls_score.float(), bbox_pred.float(), labels, valid, rois, gts,
                                                 self.num_classes, self.bbox_coder, beta, w_cls, w_bbox, pos_weight)
        return dict(loss_cls=l_cls, acc=acc, loss_bbox=l_box)


def _cfg_get(cfg, key, default=None):
    if cfg is None:
        return default
    return cfg.get(key, default) if isinstance(cfg, dict) else getattr(cfg, key, default)


@_REG.register_module()
class OrientedStandardRoIHead(nn.Module):
    """``mmrotate/models/roi_heads/oriented_standard_roi_head.py`` (on ``rotate_standard_roi_head.py:13-78``): RoI
    extractor + box head + the training path ``forward_train`` (:31-95) / ``_bbox_forward_train`` (:97-124).

    ``forward_train`` accepts the reference's ``proposal_list`` (list of (n_i, 5+) tensors) or the sync-free fixed-size
    block ``((B, P, 6) tensor, (B,) counts)`` the RPN head emits; either way assignment + sampling + targets + loss cost
    no host synchronisation: per image a masked MaxIoU assignment, ``add_gt_as_proposals``, the fixed-size sampler, then
    ONE RoI-extractor launch, the head's GEMMs and ONE fused target-encode + loss kernel for all images."""

    def __init__(self, bbox_roi_extractor=None, bbox_head=None, shared_head=None, train_cfg=None, test_cfg=None,
                 pretrained=None, init_cfg=None, version='oc'):
        super().__init__()
        if shared_head is not None:
            raise NotImplementedError('shared_head is not used by any SM3Det config')
        self.train_cfg, self.test_cfg, self.version, self.init_cfg = train_cfg, test_cfg, version, init_cfg
        self.bbox_roi_extractor = _REG.build(bbox_roi_extractor) if bbox_roi_extractor is not None else None
        self.bbox_head = _REG.build(bbox_head) if bbox_head is not None else None
        self.bbox_assigner = self.bbox_sampler = None
        if train_cfg:
            from .assign import BBOX_ASSIGNERS, BBOX_SAMPLERS
            self.bbox_assigner = BBOX_ASSIGNERS.build(train_cfg['assigner'])
            self.bbox_sampler = BBOX_SAMPLERS.build(train_cfg['sampler'])

    @property
    def with_bbox(self):
        return self.bbox_head is not None

    def init_weights(self):
        if self.bbox_head is not None:
            self.bbox_head.init_weights()

    def _bbox_forward(self, x, rois):
        """rotate_standard_roi_head.py:150-165"""
        feats = self.bbox_roi_extractor(x[:self.bbox_roi_extractor.num_inputs], rois)
        cls_score, bbox_pred = self.bbox_head(feats)
        return dict(cls_score=cls_score, bbox_pred=bbox_pred, bbox_feats=feats)

    @torch.no_grad()
    def simple_test_bboxes(self, x, img_metas, proposals, rcnn_test_cfg, rescale=False):
        """oriented_standard_roi_head.py:126-188: ONE extractor + head pass over the proposals of all images, then the box
        head's ``get_bboxes`` image by image -> (list of (k_i, 6) detections, list of (k_i,) labels)"""
        from .rpn_head import rbbox2roi
        rois = rbbox2roi(proposals)
        res = self._bbox_forward(x, rois)
        counts = [int(p.shape[0]) for p in proposals]
        rois_l, cls_l, reg_l = rois.split(counts, 0), res['cls_score'].split(counts, 0), res['bbox_pred'].split(counts, 0)
        det_bboxes, det_labels = [], []
        for i in range(len(proposals)):
            d, l = self.bbox_head.get_bboxes(rois_l[i], cls_l[i], reg_l[i], img_metas[i].get('img_shape'),
                                             img_metas[i].get('scale_factor'), rescale=rescale, cfg=rcnn_test_cfg)
            det_bboxes.append(d)
            det_labels.append(l)
        return det_bboxes, det_labels

    @torch.no_grad()
    def simple_test(self, x, proposal_list, img_metas, rescale=False):
        """rotate_standard_roi_head.py:235-262: per image a list over the classes of (k, 6) float32 arrays"""
        from .post_processing import rbbox2result
        assert self.with_bbox, 'Bbox head must be implemented.'
        det_bboxes, det_labels = self.simple_test_bboxes(x, img_metas, proposal_list, self.test_cfg, rescale=rescale)
        return [rbbox2result(det_bboxes[i], det_labels[i], self.bbox_head.num_classes) for i in range(len(det_bboxes))]

    def sample_fixed(self, proposals, counts, gt_bboxes, gt_labels, generator=None):
        """Assign + sample every image on the device.  proposals (B, P, 5+), counts (B,) or None -> dict of fixed-size
        blocks: rois (B*S, 6) [batch index | box], labels (B*S,), valid (B*S,), gts (B*S, 5), n_pos / n_neg (B,)."""
        B, P = proposals.shape[0], proposals.shape[1]
        dev = proposals.device
        S, C = self.bbox_sampler.num, self.bbox_head.num_classes
        slot = torch.arange(P, device=dev)
        rois = torch.empty(B * S, 6, device=dev)
        labels = torch.empty(B * S, dtype=torch.long, device=dev)
        gts_out = torch.empty(B * S, 5, device=dev)
        valid_out = torch.empty(B * S, dtype=torch.uint8, device=dev)
        npos, nneg = [], []
        lib = _lib.lib()
        for i in range(B):
            boxes = proposals[i]  # (P, 5+): the kernels take the row stride
            g, gl = gt_bboxes[i].float()[:, :5].contiguous(), gt_labels[i].long().contiguous()
            k = int(g.shape[0])
            flags = (slot < counts[i]).to(torch.uint8) if counts is not None else None
            ar = self.bbox_assigner.assign(boxes, g, None, gl, box_flags=flags)
            gt_inds = ar.gt_inds
            prepend = bool(self.bbox_sampler.add_gt_as_proposals and k > 0)
            if prepend:  # BaseSampler.sample: gts first, self-matched
                gt_inds = torch.cat([torch.arange(1, k + 1, device=dev), gt_inds])
            idx, is_pos, valid, n_pos, n_neg = self.bbox_sampler.sample_fixed(gt_inds, generator)
            # RoI rows (unused slots: a unit box), labels (background for negatives), matched gts: one launch
            with torch.cuda.device(dev):
                _lib.check(lib.sm3_rcnn_gather_samples(
                    _lib.ptr(g), _lib.ptr(gl), k, int(prepend), _lib.ptr(boxes), boxes.stride(0), _lib.ptr(gt_inds),
                    _lib.ptr(ar.labels), _lib.ptr(idx), _lib.ptr(is_pos), _lib.ptr(valid), S, C, float(i), i * S,
                    _lib.ptr(rois), _lib.ptr(labels), _lib.ptr(gts_out), _lib.ptr(valid_out), _lib.stream_ptr()),
                    'rcnn_gather_samples')
            npos.append(n_pos)
            nneg.append(n_neg)
        return dict(rois=rois, labels=labels, valid=valid_out.view(torch.bool), gts=gts_out, n_pos=torch.stack(npos),
                    n_neg=torch.stack(nneg))

    def forward_train(self, x, img_metas, proposal_list, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_masks=None,
                      generator=None, return_samples=False):
        if isinstance(proposal_list, (tuple, list)) and len(proposal_list) == 2 and torch.is_tensor(proposal_list[0]) \
                and proposal_list[0].dim() == 3:
            proposals, counts = proposal_list
        else:  # the reference's list of per-image (n_i, 5+) proposals: pad to one block (shapes are host-side facts)
            P = max([int(p.shape[0]) for p in proposal_list] + [1])
            proposals = proposal_list[0].new_zeros(len(proposal_list), P, 5)
            for i, p in enumerate(proposal_list):
                proposals[i, :p.shape[0]] = p[:, :5]
            counts = torch.tensor([int(p.shape[0]) for p in proposal_list], device=proposals.device)
        smp = self.sample_fixed(proposals.detach(), counts, gt_bboxes, gt_labels, generator)
        res = self._bbox_forward(x, smp['rois'])
        pw = float((self.train_cfg or {}).get('pos_weight', -1))
        losses = self.bbox_head.loss_fused(res['cls_score'], res['bbox_pred'], smp['labels'], smp['valid'],
                                           smp['rois'][:, 1:], smp['gts'], pw)
        return (losses, smp) if return_samples else losses
