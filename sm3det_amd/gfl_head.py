"""``GFLHead`` conv towers on the MI355X kernels -- SURVEY.md 8(f) row 2, the SAR branch's dense head
(``local_configs/main_SM3Det.py:29-48``: ``type='GFLHead', num_classes=26, in_channels=256, stacked_convs=4,
feat_channels=256, reg_max=16``, strides 8..128).

The class is mmdet 2.x code (``mmdet/models/dense_heads/gfl_head.py``), which the reference does NOT vendor: its
semantics are restated here from the published implementation -- **parity unpinned** (no reference source or test in
/root/reference to check against; the oracle ``oracle/gfl_oracle.py`` is the same restatement in plain torch ops):

    cls_convs / reg_convs : stacked_convs x ConvModule(3x3, pad 1, no bias, GroupNorm(32), ReLU)
    gfl_cls               : Conv2d(feat, num_classes, 3, padding=1)            (sigmoid classification)
    gfl_reg               : Conv2d(feat, 4 * (reg_max + 1), 3, padding=1)
    scales                : one learnable scalar per pyramid level (``Scale(1.0)``)
    forward_single(x, scale) -> (gfl_cls(cls_feat), scale(gfl_reg(reg_feat)).float())

``state_dict`` keys and shapes are mmdet's (``cls_convs.{i}.conv.weight`` (256,256,3,3), ``cls_convs.{i}.gn.{weight,
bias}``, ``gfl_cls.{weight,bias}``, ``gfl_reg.{weight,bias}``, ``scales.{l}.scale``).  Execution: the 3x3 convolutions
are implicit GEMMs on the MFMA GEMM family (``sm3_conv3x3_nhwc_*``, ~206 GFLOP forward per 1024^2 SAR image: the largest
block of detector FLOPs after the backbone), GroupNorm + ReLU is two HBM passes (``sm3_groupnorm_*``); everything stays
NHWC.  ``gfl_cls`` / ``gfl_reg`` run with their output channels zero-padded to a multiple of 32 (the input-gradient
GEMM contracts over them).

Round 4: the loss side -- ``forward_train`` / ``loss`` / ``get_targets`` (``ATSSAssigner(topk=9)`` + ``PseudoSampler``,
QFL / DFL / GIoU, the ``Integral`` layer) and ``get_bboxes`` / ``simple_test`` -- in plain PyTorch over fixed-shape masked
tensors (``sm3det_amd/gfl_losses.py``: same sums as mmdet's ``nonzero``-indexed form, no host sync); mmdet code as well,
restated, **parity unpinned**.
"""
import math

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib
from . import _lib_backbone as LB
from . import level_streams
from .fpn import _Conv, _to_nchw_view, _to_nhwc, conv3x3_nhwc
from .registry import MODELS as _REG

call = LB.call


class _GroupNormReLU(Function):
    """y = relu(GroupNorm(x)) on (B,H,W,C) tokens; hand-written backward (d gamma / d beta reduced on the side)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, eps, relu):
        _lib.require_gpu(x, gamma, beta)
        x = x.contiguous()
        B, H, W, C = x.shape
        P = H * W
        y = torch.empty_like(x)
        stats = torch.empty(B, groups, 2, device=x.device, dtype=torch.float32)
        nb = _lib.lib().sm3_groupnorm_workspace_bytes(B, P, C, groups)
        if nb == 0:
            raise _lib.SM3Error(f'GroupNorm({groups}) over {C} channels is not supported by the MI355X kernel')
        ws = _lib.workspace(nb, x.device)
        call('groupnorm_fwd', x, gamma.contiguous(), beta.contiguous(), float(eps), int(relu), y, stats, B, P, C,
             groups, ws, nb, nbytes=3.0 * x.numel() * 4)
        ctx.save_for_backward(x, y, gamma, stats)
        ctx.meta = (groups, relu)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, y, gamma, stats = ctx.saved_tensors
        groups, relu = ctx.meta
        B, H, W, C = x.shape
        P = H * W
        dy = dy.contiguous()
        dx = torch.empty_like(x)
        rows = B * _lib.lib().sm3_groupnorm_blocks(P, C)
        part = torch.empty(rows, 2 * C, device=x.device, dtype=torch.float32)
        nb = _lib.lib().sm3_groupnorm_workspace_bytes(B, P, C, groups)
        ws = _lib.workspace(nb, x.device)
        call('groupnorm_bwd', dy, x, y, gamma.contiguous(), stats, int(relu), dx, part, B, P, C, groups, ws, nb,
             nbytes=5.0 * x.numel() * 4)
        dgb = torch.empty(2, C, device=x.device, dtype=torch.float32)
        call('row_partials_reduce', part, rows, 2 * C, dgb)
        return dx, dgb[0], dgb[1], None, None, None


def group_norm_relu(x, gamma, beta, groups=32, eps=1e-5, relu=True):
    return _GroupNormReLU.apply(x, gamma, beta, groups, eps, relu)


class _ConvGN(nn.Module):
    """``ConvModule(conv_cfg=None, norm_cfg=GN, act=ReLU)``: children ``conv`` (bias-free) and ``gn`` as in mmcv"""

    def __init__(self, cin, cout, groups):
        super().__init__()
        self.conv = _Conv(cin, cout, 3)
        self.conv.bias = None  # ConvModule(bias='auto') drops the bias in front of a norm layer
        self.gn = nn.GroupNorm(groups, cout)

    def forward_nhwc(self, x):
        t = conv3x3_nhwc(x, self.conv.weight, None, 1, False)
        return group_norm_relu(t, self.gn.weight, self.gn.bias, self.gn.num_groups, self.gn.eps, True)


class Scale(nn.Module):
    """mmcv.cnn.Scale: a learnable scalar factor"""

    def __init__(self, scale=1.0):
        super().__init__()
        self.scale = nn.Parameter(torch.tensor(scale, dtype=torch.float))

    def forward(self, x):
        return x * self.scale


@_REG.register_module()
class GFLHead(nn.Module):
    def __init__(self, num_classes, in_channels, stacked_convs=4, feat_channels=256, conv_cfg=None,
                 norm_cfg=dict(type='GN', num_groups=32, requires_grad=True), anchor_generator=None, loss_cls=None,
                 loss_dfl=None, loss_bbox=None, bbox_coder=None, reg_max=16, train_cfg=None, test_cfg=None,
                 init_cfg=None, **kwargs):
        super().__init__()
        if conv_cfg is not None or (norm_cfg or {}).get('type') != 'GN':
            raise NotImplementedError('GFLHead on MI355X: conv_cfg None and GroupNorm towers (mmdet default) only')
        self.num_classes, self.in_channels, self.stacked_convs = num_classes, in_channels, stacked_convs
        self.feat_channels, self.reg_max = feat_channels, reg_max
        self.cls_out_channels = num_classes  # use_sigmoid_cls (QualityFocalLoss)
        ag = dict(anchor_generator or dict(type='AnchorGenerator', ratios=[1.0], octave_base_scale=8,
                                           scales_per_octave=1, strides=[8, 16, 32, 64, 128]))
        self.anchor_cfg = ag
        self.strides = list(ag['strides'])
        self.loss_cls_cfg, self.loss_dfl_cfg, self.loss_bbox_cfg = loss_cls, loss_dfl, loss_bbox
        self.train_cfg, self.test_cfg, self.init_cfg = train_cfg, test_cfg, init_cfg
        groups = int(norm_cfg.get('num_groups', 32))
        self.cls_convs, self.reg_convs = nn.ModuleList(), nn.ModuleList()
        for i in range(stacked_convs):
            chn = in_channels if i == 0 else feat_channels
            self.cls_convs.append(_ConvGN(chn, feat_channels, groups))
            self.reg_convs.append(_ConvGN(chn, feat_channels, groups))
        self.gfl_cls = _Conv(feat_channels, self.cls_out_channels, 3)
        self.gfl_reg = _Conv(feat_channels, 4 * (reg_max + 1), 3)
        self.scales = nn.ModuleList([Scale(1.0) for _ in self.strides])
        self.init_weights()

    def init_weights(self):
        """mmdet init_cfg: Normal(std=0.01) on every Conv2d, gfl_cls bias = bias_init_with_prob(0.01)"""
        for m in list(self.cls_convs) + list(self.reg_convs):
            nn.init.normal_(m.conv.weight, 0.0, 0.01)
        for m in (self.gfl_cls, self.gfl_reg):
            nn.init.normal_(m.weight, 0.0, 0.01)
            nn.init.constant_(m.bias, 0.0)
        nn.init.constant_(self.gfl_cls.bias, float(-math.log((1 - 0.01) / 0.01)))

    @staticmethod
    def _padded_conv(x, conv):
        """3x3 conv whose output channels are zero-padded to a multiple of 32 for the kernels, sliced back"""
        cout = conv.out_channels
        pad = (-cout) % 32
        w, b = conv.weight, conv.bias
        if pad:
            w = torch.cat([w, w.new_zeros(pad, 3, 3, conv.in_channels)], 0)
            b = torch.cat([b, b.new_zeros(pad)], 0)
        return conv3x3_nhwc(x, w, b, 1, False)[..., :cout]

    def forward_single(self, x, scale):
        """x logically NCHW -> (cls_score (B,num_classes,H,W), bbox_pred (B,4*(reg_max+1),H,W)) views over NHWC"""
        t = _to_nhwc(x)
        cls_feat, reg_feat = t, t
        for m in self.cls_convs:
            cls_feat = m.forward_nhwc(cls_feat)
        for m in self.reg_convs:
            reg_feat = m.forward_nhwc(reg_feat)
        cls_score = self._padded_conv(cls_feat, self.gfl_cls)
        bbox_pred = scale(self._padded_conv(reg_feat, self.gfl_reg)).float()
        return _to_nchw_view(cls_score), _to_nchw_view(bbox_pred)

    def forward(self, feats):
        """multi_apply(forward_single, feats, self.scales) -> (list of cls scores, list of bbox preds)"""
        outs = level_streams.map_levels(self.forward_single, feats, list(self.scales))
        return [o[0] for o in outs], [o[1] for o in outs]

    # ------------------------------------------------------------------------------------------- training side
    # mmdet 2.25 GFLHead.loss / loss_single / get_targets / _get_target_single (gfl_head.py:233-471) and AnchorHead.
    # get_anchors (anchor_head.py:170-211), restated over fixed-shape masked tensors -- parity unpinned (module docstring).
    def _anchors(self, featmap_sizes, device):
        """AnchorGenerator(ratios [1.0], octave_base_scale 8, scales_per_octave 1).grid_priors: ONE square anchor of side
        8 * stride per position, centred on (x * stride, y * stride) (center_offset 0); level-major list of (H*W, 4)"""
        from .rpn_head import grid_anchors
        key = (tuple(featmap_sizes), str(device))
        cache = self.__dict__.setdefault('_anchor_cache', {})
        if key in cache:  # built once per pyramid geometry: no host -> device copy on the training path (capturable)
            return cache[key]
        ag = self.anchor_cfg
        if 'octave_base_scale' in ag:
            scales = [ag['octave_base_scale'] * 2 ** (i / ag.get('scales_per_octave', 1))
                      for i in range(ag.get('scales_per_octave', 1))]
        else:
            scales = list(ag.get('scales', [8]))
        cache[key] = grid_anchors(featmap_sizes, self.strides, scales, list(ag.get('ratios', [1.0])), device=device)
        return cache[key]

    def _valid_flags(self, featmap_sizes, pad_shape, device):
        """AnchorGenerator.valid_flags (anchor_generator.py:397-450): positions inside the padded image; allowed_border -1
        (the SM3Det config) makes anchor_inside_flags == valid_flags"""
        out = []
        h, w = pad_shape[:2]
        for (fh, fw), s in zip(featmap_sizes, self.strides):
            vh, vw = min(int(math.ceil(h / s)), fh), min(int(math.ceil(w / s)), fw)
            vy = torch.arange(fh, device=device) < vh
            vx = torch.arange(fw, device=device) < vw
            out.append((vy[:, None] & vx[None, :]).reshape(-1))
        return out

    def _valid_cat(self, featmap_sizes, pad_shape, device):
        """the concatenated valid flags of one (pyramid geometry, padded image shape): built once, reused every step"""
        key = (tuple(featmap_sizes), tuple(int(v) for v in pad_shape[:2]), str(device))
        cache = self.__dict__.setdefault('_valid_cache', {})
        if key not in cache:
            cache[key] = torch.cat(self._valid_flags(featmap_sizes, pad_shape, device))
        return cache[key]

    def get_targets(self, anchors, num_level_anchors, valid_flags_list, gt_bboxes_list, gt_labels_list):
        """per image ATSS assignment + PseudoSampler -> stacked (B, A) labels (num_classes = background), label weights,
        (B, A, 4) box targets, (B, A) positive mask.  anchors (A, 4) level-major, shared by the images."""
        from . import gfl_losses as GL
        tc = self.train_cfg or {}
        asg = dict(tc.get('assigner') or dict(type='ATSSAssigner', topk=9))
        if asg.get('type') != 'ATSSAssigner':
            raise NotImplementedError(f"GFLHead on MI355X: assigner {asg.get('type')!r} (the SM3Det configs use ATSSAssigner)")
        if tc.get('allowed_border', -1) >= 0:
            raise NotImplementedError('allowed_border >= 0 (the SM3Det configs use -1)')
        pos_weight = float(tc.get('pos_weight', -1))
        labels, weights, targets, poss = [], [], [], []
        for valid, gtb, gtl in zip(valid_flags_list, gt_bboxes_list, gt_labels_list):
            gt_inds, _, _ = GL.atss_assign(anchors, num_level_anchors, gtb.float(), None, topk=int(asg.get('topk', 9)),
                                           valid=valid)
            pos = gt_inds > 0
            if gtb.size(0):
                tgt = torch.where(pos[:, None], gtb.float()[(gt_inds - 1).clamp(min=0)], anchors.new_zeros(()))
                lab = torch.where(pos, gtl.long()[(gt_inds - 1).clamp(min=0)], gt_inds.new_full((), self.num_classes))
            else:
                tgt = torch.zeros_like(anchors)
                lab = gt_inds.new_full(gt_inds.shape, self.num_classes)
            w = valid.float()  # negatives 1, positives pos_weight (<= 0: 1), anchors outside the image 0 (unmap fill)
            if pos_weight > 0:
                w = torch.where(pos, w * pos_weight, w)
            labels.append(lab)
            weights.append(w)
            targets.append(tgt)
            poss.append(pos & valid)
        return torch.stack(labels), torch.stack(weights), torch.stack(targets), torch.stack(poss)

    def loss(self, cls_scores, bbox_preds, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=None):
        """-> dict(loss_cls, loss_bbox, loss_dfl): lists with one scalar per pyramid level, as mmdet returns them"""
        from . import gfl_losses as GL
        if gt_bboxes_ignore is not None:
            raise NotImplementedError('gt_bboxes_ignore (the SM3Det detector asserts it is None)')
        sizes = [tuple(c.shape[-2:]) for c in cls_scores]
        dev = cls_scores[0].device
        lvl_anchors = self._anchors(sizes, dev)
        num_level = [int(a.shape[0]) for a in lvl_anchors]
        anchors = self._anchors_cat(sizes, lvl_anchors, dev)
        B = len(gt_bboxes)
        A, R = anchors.shape[0], self.reg_max
        tc = self.train_cfg or {}
        asg = dict(tc.get('assigner') or dict(type='ATSSAssigner', topk=9))
        if asg.get('type') != 'ATSSAssigner':
            raise NotImplementedError(f"GFLHead on MI355X: assigner {asg.get('type')!r} (the SM3Det configs use ATSSAssigner)")
        if tc.get('allowed_border', -1) >= 0:
            raise NotImplementedError('allowed_border >= 0 (the SM3Det configs use -1)')
        qb = float((self.loss_cls_cfg or {}).get('beta', 2.0))
        w_cls = float((self.loss_cls_cfg or {}).get('loss_weight', 1.0))
        w_dfl = float((self.loss_dfl_cfg or {}).get('loss_weight', 0.25))
        w_box = float((self.loss_bbox_cfg or {}).get('loss_weight', 2.0))
        valid = self._valid_u8(sizes, [m.get('pad_shape', m.get('img_shape')) for m in img_metas], dev)  # (B, A) bytes
        # mmdet runs get_targets per image (ATSSAssigner + PseudoSampler) and loss_single per level (multi_apply): ~200 torch
        # launches per step in round 5's masked torch form.  Here: one ATSS launch per image, then ONE pass per anchor of every
        # image evaluates the three losses (sums per level, the numbers loss_single returns before its normalisers) and one
        # more pass their gradients (csrc/gfl.hip); what is left in torch is the handful of scalar normalisations below.
        best = atss_best_keys(anchors, num_level, self.strides, gt_bboxes, valid, int(asg.get('topk', 9)))
        gts_all, labels_all, gt_off = self._gt_tables(gt_bboxes, gt_labels, dev)
        cs = torch.cat([c.permute(0, 2, 3, 1).reshape(B, -1, self.cls_out_channels) for c in cls_scores], 1)
        bp = torch.cat([r.permute(0, 2, 3, 1).reshape(B, -1, 4 * (R + 1)) for r in bbox_preds], 1)
        sums, wt_sum, npos = _GFLLossFn.apply(cs, bp, anchors, best, gts_all, labels_all, gt_off, valid,
                                              (tuple(num_level), tuple(self.strides), R, qb, float(tc.get('pos_weight', -1))))
        # get_targets: num_total_pos = sum over the images of max(#positives, 1); loss: reduce_mean over ranks, max(., 1)
        num_total = _reduce_mean(npos.clamp(min=1).sum().float()).clamp(min=1.0)
        avg_factor = _reduce_mean(wt_sum.sum()).clamp(min=1.0)
        l_box = sums[0] * (w_box / avg_factor)
        l_dfl = sums[1] * (w_dfl / 4.0 / avg_factor)                                        # avg_factor 4.0
        l_cls = sums[2] * (w_cls / num_total)
        return dict(loss_cls=list(l_cls.unbind(0)), loss_bbox=list(l_box.unbind(0)), loss_dfl=list(l_dfl.unbind(0)))

    def _anchors_cat(self, sizes, lvl_anchors, device):
        key = (tuple(sizes), str(device))
        cache = self.__dict__.setdefault('_anchor_cat_cache', {})
        if key not in cache:
            cache[key] = torch.cat(lvl_anchors).float().contiguous()
        return cache[key]

    def _valid_u8(self, sizes, pad_shapes, device):
        """(B, A) valid flags as bytes for one (pyramid geometry, padded shapes): built once, reused every step"""
        key = (tuple(sizes), tuple(tuple(int(v) for v in ps[:2]) for ps in pad_shapes), str(device))
        cache = self.__dict__.setdefault('_valid_u8_cache', {})
        if key not in cache:
            cache[key] = torch.stack([self._valid_cat(sizes, ps, device) for ps in pad_shapes]).to(torch.uint8).contiguous()
        return cache[key]

    def _gt_tables(self, gt_bboxes, gt_labels, device):
        """all images' gts / labels concatenated + the device vector of first-gt indices (cached per tuple of gt counts: its
        upload is a host -> device copy, which a hipGraph capture does not allow -- the warm-up step creates it)"""
        counts = tuple(int(g.shape[0]) for g in gt_bboxes)
        cache = self.__dict__.setdefault('_gt_off_cache', {})
        key = (counts, str(device))
        if key not in cache:
            off, acc = [], 0
            for c in counts:
                off.append(acc)
                acc += c
            cache[key] = torch.tensor(off, dtype=torch.int32, device=device)
        if sum(counts) == 0:
            return (torch.zeros(1, 4, device=device), torch.zeros(1, dtype=torch.long, device=device), cache[key])
        return (torch.cat([g.float().reshape(-1, 4) for g in gt_bboxes]).contiguous(),
                torch.cat([l.long().reshape(-1) for l in gt_labels]).contiguous(), cache[key])

    def loss_torch(self, cls_scores, bbox_preds, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore=None):
        """round 5's masked torch form of `loss` (get_targets + one pass over all levels in plain PyTorch): kept as the
        restatement the kernels are tested against (tests/test_gfl_gpu.py); not on the training path"""
        from . import gfl_losses as GL
        sizes = [tuple(c.shape[-2:]) for c in cls_scores]
        dev = cls_scores[0].device
        lvl_anchors = self._anchors(sizes, dev)
        num_level = [int(a.shape[0]) for a in lvl_anchors]
        anchors = torch.cat(lvl_anchors)
        valid = [self._valid_cat(sizes, m.get('pad_shape', m.get('img_shape')), dev) for m in img_metas]
        labels, label_w, box_t, pos = self.get_targets(anchors, num_level, valid, gt_bboxes, gt_labels)
        B = labels.shape[0]
        qb = float((self.loss_cls_cfg or {}).get('beta', 2.0))
        w_cls = float((self.loss_cls_cfg or {}).get('loss_weight', 1.0))
        w_dfl = float((self.loss_dfl_cfg or {}).get('loss_weight', 0.25))
        w_box = float((self.loss_bbox_cfg or {}).get('loss_weight', 2.0))
        num_total = _reduce_mean(pos.sum(dim=1).clamp(min=1).sum().float()).clamp(min=1.0)
        R = self.reg_max
        ctr, lvl1h = self._loss_tables(sizes, anchors, num_level, dev)     # (A, 2) centres / stride, (A, L) one-hot
        A = anchors.shape[0]
        cs = torch.cat([c.permute(0, 2, 3, 1).reshape(B, -1, self.cls_out_channels) for c in cls_scores], 1)
        bp = torch.cat([r.permute(0, 2, 3, 1).reshape(B, -1, 4 * (R + 1)) for r in bbox_preds], 1)
        cs, bp = cs.reshape(B * A, -1).float(), bp.reshape(B * A, -1).float()
        centers = ctr[None].expand(B, A, 2).reshape(-1, 2)
        stride = self._stride_vec[None].expand(B, A).reshape(-1, 1)
        lab, lw, ps = labels.reshape(-1), label_w.reshape(-1), pos.reshape(-1)
        psf = ps.float()
        wt = cs.detach().sigmoid().max(dim=1)[0] * psf          # weight_targets (0 off the positives)
        corners = GL.integral(bp, R)
        dec = GL.distance2bbox(centers, corners)
        dec_t = box_t.reshape(-1, 4) / stride
        score = GL.bbox_overlaps(dec.detach(), dec_t, is_aligned=True) * psf
        tgt_c = GL.bbox2distance(centers, dec_t, R).reshape(-1)
        box_a = GL.giou_loss(dec, dec_t) * wt                                                   # avg_factor 1.0
        dfl_a = (GL.distribution_focal_loss(bp.reshape(-1, R + 1), tgt_c).reshape(-1, 4) * wt[:, None]).sum(dim=1)
        cls_a = GL.quality_focal_loss(cs, lab, score, ps, qb) * lw
        per_level = torch.stack((box_a, dfl_a, cls_a, wt)).reshape(4, B, A).sum(dim=1) @ lvl1h  # (4, levels)
        avg_factor = _reduce_mean(per_level[3].sum()).clamp(min=1.0)
        l_box = per_level[0] * (w_box / avg_factor)
        l_dfl = per_level[1] * (w_dfl / 4.0 / avg_factor)                                        # avg_factor 4.0
        l_cls = per_level[2] * (w_cls / num_total)
        return dict(loss_cls=list(l_cls.unbind(0)), loss_bbox=list(l_box.unbind(0)), loss_dfl=list(l_dfl.unbind(0)))

    def _loss_tables(self, sizes, anchors, num_level, device):
        """per-anchor constants of the loss, built once per pyramid geometry (capturable: no host -> device copy later):
        anchor centres in units of the level's stride, the stride of every anchor, the (A, levels) level membership"""
        key = (tuple(sizes), str(device))
        cache = self.__dict__.setdefault('_loss_table_cache', {})
        if key not in cache:
            stride = torch.cat([anchors.new_full((n,), float(s)) for n, s in zip(num_level, self.strides)])
            ctr = torch.stack(((anchors[:, 0] + anchors[:, 2]) / 2.0, (anchors[:, 1] + anchors[:, 3]) / 2.0), dim=1) \
                / stride[:, None]
            lvl = torch.cat([torch.full((n,), i, dtype=torch.long, device=device) for i, n in enumerate(num_level)])
            cache[key] = (ctr, torch.nn.functional.one_hot(lvl, len(num_level)).float(), stride)
        ctr, onehot, self._stride_vec = cache[key]
        return ctr, onehot

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=None, proposal_cfg=None, **kwargs):
        """BaseDenseHead.forward_train (base_dense_head.py:306-345): heads -> loss"""
        cls_scores, bbox_preds = self(x)
        return self.loss(cls_scores, bbox_preds, gt_bboxes, gt_labels, img_metas, gt_bboxes_ignore)

    # ------------------------------------------------------------------------------------------- inference side
    @torch.no_grad()
    def get_bboxes(self, cls_scores, bbox_preds, img_metas, cfg=None, rescale=False):
        """GFLHead._get_bboxes_single (gfl_head.py:473-573) per image: sigmoid scores, score_thr + nms_pre top-k per level,
        Integral -> distances * stride -> boxes clipped to img_shape, then multiclass NMS (class-wise batched_nms,
        max_per_img) -> list of (dets (n, 5), labels (n,)).  Variable-length outputs: host-synchronising by nature."""
        from . import gfl_losses as GL
        from . import mmcv_ops
        cfg = dict(cfg or self.test_cfg or {})
        sizes = [tuple(c.shape[-2:]) for c in cls_scores]
        lvl_anchors = self._anchors(sizes, cls_scores[0].device)
        out = []
        for i, meta in enumerate(img_metas):
            boxes, scores, labels = [], [], []
            for cs, bp, a, stride in zip(cls_scores, bbox_preds, lvl_anchors, self.strides):
                sc = cs[i].permute(1, 2, 0).reshape(-1, self.cls_out_channels).sigmoid()
                bpi = bp[i].permute(1, 2, 0).reshape(-1, 4 * (self.reg_max + 1)).float()
                # filter_scores_and_topk: (score > thr) pairs, then the nms_pre best of them
                flat = sc.reshape(-1)
                keep = (flat > cfg.get('score_thr', 0.05)).nonzero().squeeze(1)
                k = min(int(cfg.get('nms_pre', 1000)), int(keep.numel()))
                if k == 0:
                    continue
                top = flat[keep].sort(descending=True)[1][:k]
                sel = keep[top]
                ai, ci = sel // self.cls_out_channels, sel % self.cls_out_channels
                centers = torch.stack(((a[ai, 0] + a[ai, 2]) / 2.0, (a[ai, 1] + a[ai, 3]) / 2.0), dim=1)
                bb = GL.distance2bbox(centers, GL.integral(bpi[ai], self.reg_max) * stride)
                h, w = meta['img_shape'][:2]
                bb = torch.stack((bb[:, 0].clamp(0, w), bb[:, 1].clamp(0, h), bb[:, 2].clamp(0, w), bb[:, 3].clamp(0, h)), 1)
                boxes.append(bb)
                scores.append(flat[sel])
                labels.append(ci)
            if not boxes:
                out.append((cls_scores[0].new_zeros(0, 5), cls_scores[0].new_zeros(0, dtype=torch.long)))
                continue
            bb, sc, lb = torch.cat(boxes), torch.cat(scores), torch.cat(labels)
            if rescale and meta.get('scale_factor') is not None:
                bb = bb / bb.new_tensor(meta['scale_factor'])
            dets, keep = mmcv_ops.batched_nms(bb, sc, lb, dict(cfg.get('nms', dict(type='nms', iou_threshold=0.6))))
            n = int(cfg.get('max_per_img', 100))
            out.append((dets[:n], lb[keep][:n]))
        return out

    def simple_test(self, feats, img_metas, rescale=False):
        """BBoxTestMixin.simple_test_bboxes (dense_test_mixins.py:18-38)"""
        return self.get_bboxes(*self(feats), img_metas=img_metas, rescale=rescale)



# ------------------------------------------------------------------------------------------- loss kernels (round 6)
def _level_tables(num_level, strides):
    """HOST tables of sm3_atss_assign / sm3_gfl_loss_*: level offsets and strides (ctypes arrays, built per pyramid geometry)"""
    import ctypes
    off = [0]
    for n in num_level:
        off.append(off[-1] + int(n))
    return (ctypes.c_int * len(off))(*off), (ctypes.c_float * len(num_level))(*[float(s) for s in strides])


def atss_best_keys(anchors, num_level, strides, gt_bboxes_list, valid, topk):
    """ATSS assignment of every image on the device (sm3_atss_assign, csrc/gfl.hip): (B, A) int64 keys -- 0 = no positive,
    else (IoU bits << 32 | ~gt index of the image).  valid (B, A) uint8 or None."""
    from ._lib import check, lib, ptr, require_gpu, stream_ptr
    require_gpu(anchors)
    B, A = len(gt_bboxes_list), anchors.shape[0]
    off, st = _level_tables(num_level, strides)
    best = torch.empty(B, A, dtype=torch.int64, device=anchors.device)
    anchors = anchors.float().contiguous()
    with torch.cuda.device(anchors.device):
        for b, g in enumerate(gt_bboxes_list):
            g = g.float().contiguous()
            check(lib().sm3_atss_assign(ptr(anchors), A, off, st, len(num_level), ptr(g) if g.numel() else None, int(g.shape[0]),
                                        ptr(valid[b]) if valid is not None else None, int(topk), ptr(best[b]), stream_ptr()),
                  'atss_assign')
    return best


class _GFLLossFn(Function):
    """per-level sums of the three GFL losses (GIoU x weight, DFL x weight, QFL x label weight) + weight sums + positive
    counts, one kernel forward (sm3_gfl_loss_fwd) and one backward (sm3_gfl_loss_bwd)"""

    @staticmethod
    def forward(ctx, cs, bp, anchors, best, gts, gt_labels, gt_off, valid, meta):
        from ._lib import check, lib, ptr, require_gpu, stream_ptr
        require_gpu(cs, bp, anchors, best)
        num_level, strides, R, beta, pos_weight = meta
        cs, bp = cs.float().contiguous(), bp.float().contiguous()
        B, A, C = cs.shape
        off, st = _level_tables(num_level, strides)
        sums = torch.empty(4, 8, dtype=torch.float64, device=cs.device)
        npos = torch.empty(B, dtype=torch.int32, device=cs.device)
        args = (ptr(cs), ptr(bp), ptr(anchors), B, A, C, int(R), off, st, len(num_level), ptr(best), ptr(gts), ptr(gt_labels),
                ptr(gt_off), ptr(valid) if valid is not None else None, float(pos_weight), float(beta))
        with torch.cuda.device(cs.device):
            check(lib().sm3_gfl_loss_fwd(*args, ptr(sums), ptr(npos), stream_ptr()), 'gfl_loss_fwd')
        ctx.save_for_backward(cs, bp, anchors, best, gts, gt_labels, gt_off, valid)
        ctx.meta = meta
        L = len(num_level)
        ctx.mark_non_differentiable(npos)
        out = sums[:, :L].float()
        return out[:3].contiguous(), out[3].contiguous(), npos

    @staticmethod
    def backward(ctx, d_sums, _d_wt, _d_npos):
        from ._lib import check, lib, ptr, stream_ptr
        cs, bp, anchors, best, gts, gt_labels, gt_off, valid = ctx.saved_tensors
        num_level, strides, R, beta, pos_weight = ctx.meta
        B, A, C = cs.shape
        off, st = _level_tables(num_level, strides)
        coef = torch.zeros(3, 8, dtype=torch.float32, device=cs.device)
        coef[:, :len(num_level)] = d_sums.float()
        dcs, dbp = torch.empty_like(cs), torch.empty_like(bp)
        with torch.cuda.device(cs.device):
            check(lib().sm3_gfl_loss_bwd(ptr(cs), ptr(bp), ptr(anchors), B, A, C, int(R), off, st, len(num_level), ptr(best),
                                         ptr(gts), ptr(gt_labels), ptr(gt_off), ptr(valid) if valid is not None else None,
                                         float(pos_weight), float(beta), ptr(coef), ptr(dcs), ptr(dbp), stream_ptr()),
                  'gfl_loss_bwd')
        return dcs, dbp, None, None, None, None, None, None, None


def _reduce_mean(t):
    """mmdet.core.utils.reduce_mean: mean over the ranks of a data-parallel job (identity for one process)"""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return t
    t = t.clone()
    dist.all_reduce(t.div_(dist.get_world_size()), op=dist.ReduceOp.SUM)
    return t
