"""Oriented-RPN head on the MI355X kernels: the conv tower and the proposal glue.

Mirrors, for the part that is in the reference tree:

* ``OrientedRPNHead._init_layers`` / ``RotatedRPNHead.forward_single`` (``mmrotate/models/dense_heads/
  oriented_rpn_head.py:18-24``, ``rotated_rpn_head.py:43-50``): ``rpn_conv`` 3x3 + ReLU, ``rpn_cls`` / ``rpn_reg`` 1x1 --
  same parameter names and shapes in ``state_dict``; executed on NHWC tokens: one implicit-GEMM 3x3 with a fused
  bias + ReLU epilogue, then ONE GEMM for both 1x1 heads (their weights are concatenated per call);
* ``OrientedRPNHead._get_bboxes_single`` (``oriented_rpn_head.py:189-281``): per level sigmoid -> descending sort ->
  top ``nms_pre`` -> ``MidpointOffsetCoder.decode`` -> ``obb2xyxy`` -> ``batched_nms`` over levels -> top
  ``max_per_img``; here the gather + decode + obb2xyxy is one kernel per level, sort / NMS are the library's own
  device kernels, and there is ONE host sync per image (the variable-length result) instead of one per op;
* ``MidpointOffsetCoder.decode`` (``mmrotate/core/bbox/coder/delta_midpointoffset_rbbox_coder.py:53-84``).

* ``OrientedRPNHead._get_targets_single`` / ``loss_single`` and ``RotatedRPNHead.get_targets`` / ``loss``
  (``oriented_rpn_head.py:26-187``, ``rotated_rpn_head.py:152-372``) as ``loss()`` / ``forward_train()``: masked MaxIoU
  assignment (``anchor_inside_flags`` as a per-anchor flag instead of a compaction), the sync-free sampler and ONE fused
  target-encode + loss kernel over the sampled anchors (``sm3det_amd/det_losses.py``).

mmdet 2.x pieces the reference does not vendor are restated and flagged: ``grid_anchors`` / ``valid_flags`` /
``anchor_inside_flags`` (``AnchorGenerator``, [memory]), the assigner / sampler rules (``assign.py``) and the loss
formulas (``CrossEntropyLoss(use_sigmoid)``, ``SmoothL1Loss``).  No CPU fallback.
"""
import torch
import torch.nn as nn

from . import _lib
from . import _lib_backbone as LB
from . import level_streams
from . import backbone_ops as ops
from . import mmcv_ops
from .fpn import _Conv, _to_nhwc, conv3x3_nhwc
from .registry import ROTATED_NECKS as _REG  # every ROTATED_* registry of the reference is the same MODELS object

call = LB.call


class MidpointOffsetCoder:
    """decode-only mirror of the reference coder (angle version 'le90', the one SM3Det uses)."""

    def __init__(self, target_means=(0., 0., 0., 0., 0., 0.), target_stds=(1., 1., 1., 1., 1., 1.),
                 angle_range='oc'):
        self.means = tuple(float(v) for v in target_means)
        self.stds = tuple(float(v) for v in target_stds)
        self.version = angle_range
        import ctypes
        self._means = (ctypes.c_float * 6)(*self.means)
        self._stds = (ctypes.c_float * 6)(*self.stds)

    def encode(self, bboxes, gt_bboxes):
        """reference :33-51 -> bbox2delta (:87-148): (N,4) proposals + (N,5) gt boxes -> (N,6) regression targets"""
        if self.version != 'le90':
            raise NotImplementedError("only angle version 'le90' (every SM3Det config) is implemented")
        _lib.require_gpu(bboxes, gt_bboxes)
        assert bboxes.size(0) == gt_bboxes.size(0) and bboxes.size(-1) == 4 and gt_bboxes.size(-1) == 5
        props, gt = bboxes.float().contiguous(), gt_bboxes.float().contiguous()
        out = torch.empty(props.shape[0], 6, device=props.device)
        _lib.check(_lib.lib().sm3_midpoint_offset_encode_le90(LB._p(props), LB._p(gt), props.shape[0], self._means,
                                                              self._stds, LB._p(out), _lib.stream_ptr()),
                   'midpoint_offset_encode_le90')
        return out

    def decode(self, bboxes, pred_bboxes, max_shape=None, wh_ratio_clip=16 / 1000, order=None, scores=None):
        """(N,4) anchors + (N,6) deltas -> (n,5) oriented boxes; with ``order`` (int64 permutation prefix) only the
        listed rows are decoded, in that order.  Returns (proposals, hboxes[, gathered scores])."""
        if self.version != 'le90':
            raise NotImplementedError("only angle version 'le90' (every SM3Det config) is implemented")
        _lib.require_gpu(bboxes, pred_bboxes)
        assert pred_bboxes.size(0) == bboxes.size(0) and bboxes.size(-1) == 4 and pred_bboxes.size(-1) == 6
        anchors = bboxes.float().contiguous()
        deltas = pred_bboxes.float().contiguous()
        n = int(order.numel()) if order is not None else anchors.shape[0]
        props = torch.empty(n, 5, device=anchors.device)
        hb = torch.empty(n, 4, device=anchors.device)
        so = torch.empty(n, device=anchors.device) if scores is not None else None
        L = _lib.lib()
        _lib.check(L.sm3_rpn_decode_le90(LB._p(anchors), LB._p(deltas), LB._p(scores),
                                         LB._p(order), n, self._means, self._stds, float(wh_ratio_clip),
                                         LB._p(props), LB._p(hb), LB._p(so), _lib.stream_ptr()), 'rpn_decode_le90')
        return (props, hb) if scores is None else (props, hb, so)


class DeltaXYWHAOBBoxCoder:
    """Mirror of ``mmrotate/core/bbox/coder/delta_xywha_rbbox_coder.py`` (class :11-108, ``bbox2delta`` :112-176,
    ``delta2bbox`` :180-283) for angle range 'le90', class-agnostic (N,5) deltas and ``add_ctr_clamp=False`` -- the RoI
    head's coder in main_SM3Det.py (edge_swap=True, proj_xy=True, stds (0.1,0.1,0.2,0.2,0.1))."""

    def __init__(self, target_means=(0., 0., 0., 0., 0.), target_stds=(1., 1., 1., 1., 1.), angle_range='oc',
                 norm_factor=None, edge_swap=False, proj_xy=False, add_ctr_clamp=False, ctr_clamp=32):
        import ctypes
        if angle_range != 'le90':
            raise NotImplementedError("only angle_range 'le90' (every SM3Det config) is implemented")
        if add_ctr_clamp:
            raise NotImplementedError('add_ctr_clamp (YOLOF only) is not implemented')
        self.means, self.stds = tuple(float(v) for v in target_means), tuple(float(v) for v in target_stds)
        self.angle_range, self.norm_factor = angle_range, norm_factor
        self.edge_swap, self.proj_xy = bool(edge_swap), bool(proj_xy)
        self.add_ctr_clamp, self.ctr_clamp = add_ctr_clamp, ctr_clamp
        self._means = (ctypes.c_float * 5)(*self.means)
        self._stds = (ctypes.c_float * 5)(*self.stds)

    def encode(self, bboxes, gt_bboxes):
        _lib.require_gpu(bboxes, gt_bboxes)
        assert bboxes.size(0) == gt_bboxes.size(0) and bboxes.size(-1) == 5 and gt_bboxes.size(-1) == 5
        props, gt = bboxes.float().contiguous(), gt_bboxes.float().contiguous()
        out = torch.empty_like(props)
        _lib.check(_lib.lib().sm3_delta_xywha_encode_le90(LB._p(props), LB._p(gt), props.shape[0], self._means,
                                                          self._stds, float(self.norm_factor or 0.0),
                                                          int(self.edge_swap), int(self.proj_xy), LB._p(out),
                                                          _lib.stream_ptr()), 'delta_xywha_encode_le90')
        return out

    def decode(self, bboxes, pred_bboxes, max_shape=None, wh_ratio_clip=16 / 1000):
        _lib.require_gpu(bboxes, pred_bboxes)
        assert pred_bboxes.size(0) == bboxes.size(0)
        if pred_bboxes.size(-1) != 5 or bboxes.size(-1) != 5:
            raise NotImplementedError('per-class deltas (reg_class_agnostic=False) are not used by the SM3Det config')
        rois, deltas = bboxes.float().contiguous(), pred_bboxes.float().contiguous()
        out = torch.empty_like(rois)
        mh, mw = (int(max_shape[0]), int(max_shape[1])) if max_shape is not None else (0, 0)
        _lib.check(_lib.lib().sm3_delta_xywha_decode_le90(LB._p(rois), LB._p(deltas), rois.shape[0], self._means,
                                                          self._stds, float(wh_ratio_clip),
                                                          float(self.norm_factor or 0.0), int(self.edge_swap),
                                                          int(self.proj_xy), mh, mw, LB._p(out), _lib.stream_ptr()),
                   'delta_xywha_decode_le90')
        return out


def rbbox2roi(bbox_list):
    """mmrotate/core/bbox/transforms.py rbbox2roi: list of per-image (n,5+) boxes -> (sum n, 6) [batch_ind, box]"""
    rois = []
    for i, b in enumerate(bbox_list):
        if b.size(0) > 0:
            rois.append(torch.cat([b.new_full((b.size(0), 1), i), b[:, :5]], dim=-1))
        else:
            rois.append(b.new_zeros((0, 6)))
    return torch.cat(rois, 0)


def _top_order(scores, n, k):
    """indices of the k highest of `scores` (n,) in descending score order, ties by lower index = the first k entries of
    the descending sort the reference takes (oriented_rpn_head.py:248-254).  k <= 2048: the merge-tree top-k (one LDS sort
    per 4096 scores + log2 merges) instead of a full sort of up to 196 608 scores."""
    L = _lib.lib()
    if k <= 2048:
        nb = L.sm3_topk_desc_workspace_bytes(n)
        ws = _lib.workspace(nb, scores.device)
        order = torch.empty(k, dtype=torch.int64, device=scores.device)
        call('topk_desc_f32', scores, n, k, order, ws, nb)
        return order
    nb = L.sm3_argsort_desc_workspace_bytes(n)
    ws = _lib.workspace(nb, scores.device)
    full = torch.empty(n, dtype=torch.int64, device=scores.device)
    call('argsort_desc_f32', scores, n, full, ws, nb)
    return full[:k]


def grid_anchors(featmap_sizes, strides, scales=(8,), ratios=(0.5, 1.0, 2.0), device='cuda'):
    """[memory] mmdet ``AnchorGenerator(scales, ratios, strides).grid_priors``: per level (H*W*A, 4) x1,y1,x2,y2,
    position-major (y, x) then base anchor (ratio-major, scale-minor), centre offset 0."""
    out = []
    for (H, W), s in zip(featmap_sizes, strides):
        r = torch.tensor(ratios, dtype=torch.float32)
        sc = torch.tensor(scales, dtype=torch.float32)
        hr, wr = torch.sqrt(r), 1.0 / torch.sqrt(r)
        ws = (s * wr[:, None] * sc[None, :]).reshape(-1)
        hs = (s * hr[:, None] * sc[None, :]).reshape(-1)
        base = torch.stack([-0.5 * ws, -0.5 * hs, 0.5 * ws, 0.5 * hs], -1)
        sx = torch.arange(W, dtype=torch.float32) * s
        sy = torch.arange(H, dtype=torch.float32) * s
        yy, xx = torch.meshgrid(sy, sx, indexing='ij')
        shifts = torch.stack([xx.reshape(-1), yy.reshape(-1), xx.reshape(-1), yy.reshape(-1)], -1)
        out.append((shifts[:, None, :] + base[None, :, :]).reshape(-1, 4).to(device))
    return out


class _SplitClsReg(torch.autograd.Function):
    """(B, H, W, NP) fused head output -> the (B, A, H, W) objectness and (B, 6A, H, W) delta maps as views of it.  Plain
    slicing does the same forward; its backward, however, builds a zero map per slice, copies the slice gradient in and adds
    the two (5 launches over maps of up to 17 MB per level).  The fused RPN loss writes both gradients into ONE zero-filled
    buffer of the fused layout (det_losses._RPNLoss.backward), so when the two incoming gradients are the matching views of
    one such buffer it IS the gradient of the fused output and is returned as it stands."""

    fast_path, fast_hits = True, 0  # (test hooks: tests/test_rpn_gpu.py checks the shortcut is taken and changes nothing)

    @staticmethod
    def forward(ctx, o, A):
        ctx.A, ctx.shape = A, tuple(o.shape)
        return o[..., :A].permute(0, 3, 1, 2), o[..., A:7 * A].permute(0, 3, 1, 2)

    @staticmethod
    def backward(ctx, g_cls, g_reg):
        A, shape = ctx.A, ctx.shape
        base = None if g_cls is None or g_reg is None else g_cls._base
        if (base is not None and base is g_reg._base and tuple(base.shape) == shape and base.is_contiguous()
                and g_cls.storage_offset() == base.storage_offset()
                and g_reg.storage_offset() == base.storage_offset() + A
                and g_cls.stride() == (base.stride(0), 1, base.stride(1), base.stride(2))
                and g_reg.stride() == g_cls.stride() and getattr(base, '_sm3_rest_is_zero', False)
                and _SplitClsReg.fast_path):
            _SplitClsReg.fast_hits += 1
            return base, None
        ref = g_cls if g_cls is not None else g_reg
        g = ref.new_zeros(shape)
        if g_cls is not None:
            g[..., :A] = g_cls.permute(0, 2, 3, 1)
        if g_reg is not None:
            g[..., A:7 * A] = g_reg.permute(0, 2, 3, 1)
        return g, None


@_REG.register_module()
class OrientedRPNHead(nn.Module):
    def __init__(self, in_channels, feat_channels=256, version='oc', anchor_generator=None, bbox_coder=None,
                 loss_cls=None, loss_bbox=None, train_cfg=None, test_cfg=None,
                 init_cfg=dict(type='Normal', layer='Conv2d', std=0.01), **kwargs):
        super().__init__()
        self.in_channels, self.feat_channels, self.version = in_channels, feat_channels, version
        ag = dict(anchor_generator or dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0],
                                           strides=[4, 8, 16, 32, 64]))
        self.anchor_cfg = ag
        self.num_anchors = len(ag.get('ratios', [1.0])) * len(ag.get('scales', [8]))
        self.use_sigmoid_cls = True if loss_cls is None else bool(loss_cls.get('use_sigmoid', False))
        if not self.use_sigmoid_cls:
            raise NotImplementedError('softmax RPN classification is not used by any SM3Det config')
        self.cls_out_channels = 1
        bc = dict(bbox_coder or dict(type='MidpointOffsetCoder', angle_range=version))
        bc.pop('type', None)
        self.bbox_coder = MidpointOffsetCoder(**bc)
        self.loss_cls_cfg = dict(loss_cls or dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0))
        self.loss_bbox_cfg = dict(loss_bbox or dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0))
        if self.loss_cls_cfg.get('type', 'CrossEntropyLoss') != 'CrossEntropyLoss' or \
                self.loss_bbox_cfg.get('type', 'SmoothL1Loss') != 'SmoothL1Loss':
            raise NotImplementedError('the RPN losses of the SM3Det configs are CrossEntropyLoss(use_sigmoid) + SmoothL1Loss')
        self.train_cfg, self.test_cfg, self.init_cfg = train_cfg, test_cfg, init_cfg
        self.num_classes, self.reg_decoded_bbox, self.sampling = 1, False, True  # RotatedRPNHead / AnchorHead defaults
        self.assigner = self.sampler = None
        if train_cfg is not None:
            from .assign import BBOX_ASSIGNERS, BBOX_SAMPLERS
            self.assigner = BBOX_ASSIGNERS.build(train_cfg['assigner'])
            if 'sampler' not in train_cfg:
                # mmdet's AnchorHead falls back to PseudoSampler (every anchor is a sample) when the config names no
                # sampler; every SM3Det config names RandomSampler(num=256, pos_fraction=0.5).  Not guessed here.
                raise NotImplementedError('OrientedRPNHead: train_cfg without `sampler` (mmdet would use PseudoSampler)')
            if train_cfg['sampler'].get('add_gt_as_proposals', False):
                # mmdet's RandomSampler defaults this flag to True; the SM3Det rpn configs set it to False
                # (local_configs/main_SM3Det.py:174) and the anchor loss below does not implement the True form
                raise NotImplementedError('OrientedRPNHead: rpn sampler with add_gt_as_proposals=True')
            self.sampler = BBOX_SAMPLERS.build(train_cfg['sampler'])
        self._anchor_cache = {}
        self._ids_cache = {}
        self._init_layers()

    def _init_layers(self):
        """reference oriented_rpn_head.py:18-24 (parameters named rpn_conv / rpn_cls / rpn_reg .weight/.bias)"""
        self.rpn_conv = _Conv(self.in_channels, self.feat_channels, 3)
        self.rpn_cls = _Conv(self.feat_channels, self.num_anchors * self.cls_out_channels, 1)
        self.rpn_reg = _Conv(self.feat_channels, self.num_anchors * 6, 1)

    def init_weights(self):
        std = float((self.init_cfg or {}).get('std', 0.01))
        for m in (self.rpn_conv, self.rpn_cls, self.rpn_reg):
            nn.init.normal_(m.weight, 0.0, std)
            nn.init.constant_(m.bias, 0.0)

    # ------------------------------------------------------------------------------------------ conv tower
    def _fused_heads(self):
        """[rpn_cls; rpn_reg; 0] as one (NP, feat) matrix, NP = rows padded to a multiple of 32 (the input-gradient
        GEMM contracts over them: K granule of the fp32 MFMA tiles)."""
        A = self.num_anchors
        n = 7 * A
        pad = (-n) % 32
        w = torch.cat([self.rpn_cls.weight, self.rpn_reg.weight,
                       self.rpn_reg.weight.new_zeros(pad, self.feat_channels)], 0)
        b = torch.cat([self.rpn_cls.bias, self.rpn_reg.bias, self.rpn_reg.bias.new_zeros(pad)], 0)
        return w, b

    def forward_single(self, x, fused=None):
        """reference rotated_rpn_head.py:43-50; x logically NCHW -> (cls (B,A,H,W), reg (B,6A,H,W)) views"""
        w, b = fused if fused is not None else self._fused_heads()
        A = self.num_anchors
        t = conv3x3_nhwc(_to_nhwc(x), self.rpn_conv.weight, self.rpn_conv.bias, 1, True)  # conv + bias + ReLU
        B, H, W, C = t.shape
        o = ops.linear(t.reshape(-1, C), w, b).view(B, H, W, -1)
        return _SplitClsReg.apply(o, A)

    def forward(self, feats):
        """multi_apply(forward_single, feats) -> (list of cls scores, list of bbox preds)"""
        fused = self._fused_heads()
        outs = level_streams.map_levels(lambda f: self.forward_single(f, fused), feats)
        return [o[0] for o in outs], [o[1] for o in outs]

    # ------------------------------------------------------------------------------------------ proposals
    def _get_bboxes_single(self, cls_scores, bbox_preds, mlvl_anchors, img_shape=None, scale_factor=None, cfg=None,
                           rescale=False):
        """reference oriented_rpn_head.py:189-281.  cls_scores[l] (A,H,W), bbox_preds[l] (6A,H,W), mlvl_anchors[l]
        (H*W*A, 4).  Returns (n, 6): cx, cy, w, h, a, score."""
        cfg = dict(self.test_cfg if cfg is None else cfg)
        nms_pre, max_per_img = int(cfg.get('nms_pre', -1)), int(cfg['max_per_img'])
        min_size = float(cfg.get('min_bbox_size', 0))
        L = _lib.lib()
        props, hboxes, scores, ids = [], [], [], []
        for idx in range(len(cls_scores)):
            assert cls_scores[idx].shape[-2:] == bbox_preds[idx].shape[-2:]
            logits = cls_scores[idx].permute(1, 2, 0).reshape(-1).float().contiguous()
            deltas = bbox_preds[idx].permute(1, 2, 0).reshape(-1, 6)
            n = logits.numel()
            sc = torch.empty_like(logits)
            call('sigmoid_f32', logits, sc, n)
            order = None
            if nms_pre > 0 and n > nms_pre:  # sort descending, keep the first nms_pre (:248-254)
                order = _top_order(sc, n, nms_pre)
            p, hb, s = self.bbox_coder.decode(mlvl_anchors[idx], deltas, max_shape=img_shape, order=order,
                                              scores=sc)
            props.append(p)
            hboxes.append(hb)
            scores.append(s)
            ids.append(torch.full((s.numel(),), idx, dtype=torch.long, device=s.device))
        proposals, hproposals = torch.cat(props), torch.cat(hboxes)
        scores, ids = torch.cat(scores), torch.cat(ids)
        if min_size > 0:
            valid = (proposals[:, 2] >= min_size) & (proposals[:, 3] >= min_size)
            proposals, hproposals, scores, ids = proposals[valid], hproposals[valid], scores[valid], ids[valid]
        if proposals.numel() == 0:
            return proposals.new_zeros(0, 5)
        _, keep = mmcv_ops.batched_nms(hproposals, scores, ids, cfg['nms'])
        dets = torch.cat([proposals, scores[:, None]], dim=1)[keep]
        return dets[:max_per_img]

    def get_bboxes_fixed(self, cls_scores, bbox_preds, img_shape=None, cfg=None, mlvl_anchors=None):
        """Sync-free proposal generation for the training step: the same pipeline as ``_get_bboxes_single`` (per level
        sigmoid -> top nms_pre -> MidpointOffsetCoder.decode -> obb2xyxy, then level-aware NMS on the horizontal
        boxes), but the result is a FIXED-size block per image, (B, max_per_img, 6) proposals (cx,cy,w,h,a,score) +
        (B,) valid counts on the device; slots past the count are zero boxes.  Requires min_bbox_size == 0 and every
        level to have >= nms_pre candidates or all of them used (true for the SM3Det configs), so no shape depends on
        data and the whole thing can sit inside a hipGraph."""
        from . import mmcv_ext
        cfg = dict(self.test_cfg if cfg is None else cfg)
        nms_pre, max_per_img = int(cfg.get('nms_pre', -1)), int(cfg['max_per_img'])
        if float(cfg.get('min_bbox_size', 0)) > 0:
            raise NotImplementedError('get_bboxes_fixed: min_bbox_size > 0 makes the proposal count data dependent')
        thr = float(cfg['nms'].get('iou_threshold', cfg['nms'].get('iou_thr', 0.7)))
        B = cls_scores[0].shape[0]
        if mlvl_anchors is None:
            sizes = [tuple(c.shape[-2:]) for c in cls_scores]
            mlvl_anchors = grid_anchors(sizes, self.anchor_cfg['strides'], self.anchor_cfg.get('scales', [8]),
                                        self.anchor_cfg.get('ratios', [1.0]), device=cls_scores[0].device)
        L = _lib.lib()
        out = cls_scores[0].new_zeros(B, max_per_img, 6)
        counts = torch.zeros(B, dtype=torch.int32, device=out.device)
        slot = torch.arange(max_per_img, device=out.device)
        for i in range(B):
            props, hboxes, scores, ids = [], [], [], []
            for idx in range(len(cls_scores)):
                logits = cls_scores[idx][i].detach().permute(1, 2, 0).reshape(-1).float().contiguous()
                deltas = bbox_preds[idx][i].detach().permute(1, 2, 0).reshape(-1, 6)
                n = logits.numel()
                sc = torch.empty_like(logits)
                call('sigmoid_f32', logits, sc, n)
                order = None
                if nms_pre > 0 and n > nms_pre:
                    order = _top_order(sc, n, nms_pre)
                p, hb, s = self.bbox_coder.decode(mlvl_anchors[idx], deltas, max_shape=img_shape, order=order, scores=sc)
                props.append(p)
                hboxes.append(hb)
                scores.append(s)
                ids.append(int(s.numel()))
            proposals, hprop = torch.cat(props), torch.cat(hboxes)
            scores, ids = torch.cat(scores), self._level_ids(tuple(ids), scores[0].device)
            # batched_nms offset trick (mmcv/ops/nms.py:300-304), then ONE NMS whose keep count stays on the device
            off = ids * (hprop.max() + 1.0)
            keep, num = mmcv_ext.nms_fixed((hprop + off[:, None]).contiguous(), scores.contiguous(), thr, 0)
            m = min(max_per_img, keep.numel())
            cnt = torch.clamp(num[0], max=m)
            valid = slot[:m] < cnt
            kidx = torch.where(valid, keep[:m], torch.zeros_like(keep[:m]))
            dets = torch.cat([proposals, scores[:, None]], 1)[kidx] * valid[:, None].to(proposals.dtype)
            out[i, :m] = dets
            counts[i] = cnt
        return out, counts

    def _level_ids(self, counts, device):
        """the level index of every concatenated candidate as a float vector (batched_nms's `idxs`): a function of the
        per-level candidate counts only, so it is built once per configuration instead of with 6 launches per image"""
        key = (counts, str(device))
        hit = self._ids_cache.get(key)
        if hit is None:
            hit = self._ids_cache[key] = torch.cat([torch.full((n,), float(i), dtype=torch.float32, device=device)
                                                    for i, n in enumerate(counts)])
        return hit

    # ------------------------------------------------------------------------------------------ targets + loss
    def _train_anchors(self, sizes, pad_shape, img_shape, device):
        """(levels' anchors, concatenated anchors, inside flags uint8) for one (feature sizes, image shape): built once
        and cached -- [memory] mmdet AnchorGenerator.grid_priors / valid_flags(pad_shape) / anchor_inside_flags(
        img_shape, train_cfg.allowed_border), the inputs of `_get_targets_single` (oriented_rpn_head.py:63-69)."""
        key = (tuple(sizes), tuple(pad_shape[:2]), tuple(img_shape[:2]), str(device))
        hit = self._anchor_cache.get(key)
        if hit is not None:
            return hit
        strides = self.anchor_cfg['strides']
        lvl = grid_anchors(sizes, strides, self.anchor_cfg.get('scales', [8]), self.anchor_cfg.get('ratios', [1.0]),
                           device='cpu')
        flags = []
        for (H, W), s in zip(sizes, strides):  # valid_flags: positions whose cell starts inside the padded image
            vh, vw = min(-(-int(pad_shape[0]) // s), H), min(-(-int(pad_shape[1]) // s), W)
            v = torch.zeros(H, W, dtype=torch.bool)
            v[:vh, :vw] = True
            flags.append(v.reshape(-1, 1).expand(H * W, self.num_anchors).reshape(-1))
        flat, valid = torch.cat(lvl), torch.cat(flags)
        border = (self.train_cfg or {}).get('allowed_border', 0)
        if border >= 0:
            ih, iw = int(img_shape[0]), int(img_shape[1])
            valid = valid & (flat[:, 0] >= -border) & (flat[:, 1] >= -border) & (flat[:, 2] < iw + border) & \
                (flat[:, 3] < ih + border)
        out = ([a.to(device) for a in lvl], flat.to(device), valid.to(torch.uint8).to(device))
        self._anchor_cache[key] = out
        return out

    def loss(self, cls_scores, bbox_preds, gt_bboxes, img_metas, gt_bboxes_ignore=None, generator=None,
             return_samples=False):
        """RotatedRPNHead.loss (rotated_rpn_head.py:305-372) with OrientedRPNHead's targets (oriented_rpn_head.py:26-134)
        and loss_single (:136-187): per image masked MaxIoU assignment of the anchors against obb2xyxy(gt) and the
        sync-free sampler; then ONE kernel over the sampled anchors of all images and levels.  Returns
        dict(loss_rpn_cls=[per level], loss_rpn_bbox=[per level]) like the reference.
        gt_bboxes: list of (k_i, 5) oriented boxes per image (on the GPU); img_metas: list of dicts with 'img_shape'
        (and optionally 'pad_shape')."""
        from . import det_losses
        if self.assigner is None:
            raise RuntimeError('OrientedRPNHead.loss needs train_cfg (assigner / sampler)')
        if gt_bboxes_ignore is not None and any(g is not None and g.numel() for g in gt_bboxes_ignore):
            raise NotImplementedError('gt_bboxes_ignore is not used by any SM3Det config')
        B = cls_scores[0].shape[0]
        dev = cls_scores[0].device
        sizes = [tuple(c.shape[-2:]) for c in cls_scores]
        idxs, poss, vals, gtis, npos, nneg = [], [], [], [], [], []
        kmax = max([int(g.shape[0]) for g in gt_bboxes] + [1])
        gts = torch.zeros(B, kmax, 5, device=dev)
        for i in range(B):
            meta = img_metas[i]
            shape = meta['img_shape']
            _, flat, inside = self._train_anchors(sizes, meta.get('pad_shape', shape), shape, dev)
            g = gt_bboxes[i].float()
            k = int(g.shape[0])
            if k:
                gts[i, :k] = g[:, :5]
            ar = self.assigner.assign(flat, det_losses.obb2xyxy(g, self.version) if k else g.new_zeros(0, 4), None, None,
                                      box_flags=inside)
            idx, is_pos, valid, n_pos, n_neg = self.sampler.sample_fixed(ar.gt_inds, generator)
            idxs.append(idx); poss.append(is_pos); vals.append(valid); gtis.append(ar.gt_inds)
            npos.append(n_pos); nneg.append(n_neg)
        idx, is_pos, valid = torch.stack(idxs), torch.stack(poss), torch.stack(vals)
        gt_inds, n_pos, n_neg = torch.stack(gtis), torch.stack(npos), torch.stack(nneg)
        pw = float((self.train_cfg or {}).get('pos_weight', -1))
        l_cls, l_box = det_losses.rpn_loss(
            [c.float() for c in cls_scores], [r.float() for r in bbox_preds], flat, idx, is_pos, valid, gt_inds, gts,
            n_pos, n_neg, self.num_anchors, self.bbox_coder.means, self.bbox_coder.stds,
            beta=float(self.loss_bbox_cfg.get('beta', 1.0)), loss_weight_cls=float(self.loss_cls_cfg.get('loss_weight', 1.0)),
            loss_weight_bbox=float(self.loss_bbox_cfg.get('loss_weight', 1.0)), pos_weight=pw)
        losses = dict(loss_rpn_cls=list(l_cls.unbind(0)), loss_rpn_bbox=list(l_box.unbind(0)))
        if return_samples:
            return losses, dict(idx=idx, is_pos=is_pos, valid=valid, gt_inds=gt_inds, n_pos=n_pos, n_neg=n_neg,
                                anchors=flat, inside=inside)
        return losses

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=None, proposal_cfg=None,
                      fixed_size=True, generator=None):
        """mmdet BaseDenseHead.forward_train as RotatedRPNHead inherits it: heads -> loss -> proposals.  Returns
        (losses, proposals): with ``fixed_size`` (default) proposals = ((B, max_per_img, 6) tensor, (B,) counts), sync-free;
        otherwise the reference's list of (n_i, 6) tensors (one host sync per image)."""
        cls_scores, bbox_preds = self(x)
        losses = self.loss(cls_scores, bbox_preds, gt_bboxes, img_metas, gt_bboxes_ignore, generator=generator)
        if proposal_cfg is None:
            return losses
        with torch.no_grad():
            # the fixed-size path decodes the whole batch against ONE image extent (one anchor set, one clamp); with
            # keep-ratio resize + padding the images of a batch may differ: such a batch takes the per-image path (each
            # image decoded with its own img_meta['img_shape'], as the reference does) instead of clamping image i > 0 to
            # image 0's extent.  The RoI head accepts either form.
            mixed = any(tuple(m['img_shape'][:2]) != tuple(img_metas[0]['img_shape'][:2]) or
                        tuple(m.get('pad_shape', m['img_shape'])[:2]) != tuple(img_metas[0].get('pad_shape', img_metas[0]['img_shape'])[:2])
                        for m in img_metas)
            if fixed_size and not mixed:
                shape = img_metas[0]['img_shape']
                lvl, _, _ = self._train_anchors([tuple(c.shape[-2:]) for c in cls_scores],
                                                img_metas[0].get('pad_shape', shape), shape, cls_scores[0].device)
                props = self.get_bboxes_fixed(cls_scores, bbox_preds, shape, proposal_cfg, mlvl_anchors=lvl)
            else:
                props = self.get_bboxes(cls_scores, bbox_preds, img_metas, proposal_cfg)
        return losses, props

    @torch.no_grad()
    def simple_test_rpn(self, x, img_metas):
        """mmdet RPNTestMixin.simple_test_rpn: the head's outputs -> proposals under ``self.test_cfg``"""
        return self.get_bboxes(*self(x), img_metas=img_metas)

    def get_bboxes(self, cls_scores, bbox_preds, img_metas=None, cfg=None, rescale=False, mlvl_anchors=None):
        """per-image proposals (list of (n,6) tensors) from the multi-level head outputs"""
        num_imgs = cls_scores[0].shape[0]
        if mlvl_anchors is None:
            sizes = [tuple(c.shape[-2:]) for c in cls_scores]
            mlvl_anchors = grid_anchors(sizes, self.anchor_cfg['strides'], self.anchor_cfg.get('scales', [8]),
                                        self.anchor_cfg.get('ratios', [1.0]), device=cls_scores[0].device)
        out = []
        for i in range(num_imgs):
            shape = None if img_metas is None else img_metas[i].get('img_shape')
            out.append(self._get_bboxes_single([c[i].detach() for c in cls_scores],
                                               [r[i].detach() for r in bbox_preds], mlvl_anchors, shape, None, cfg,
                                               rescale))
        return out
