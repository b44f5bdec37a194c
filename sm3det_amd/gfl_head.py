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
GEMM contracts over them).  Losses / ATSS assignment / the Integral decoder are mmdet code as well and not built.
"""
import math

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib
from . import _lib_backbone as LB
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
        outs = [self.forward_single(f, s) for f, s in zip(feats, self.scales)]
        return [o[0] for o in outs], [o[1] for o in outs]
