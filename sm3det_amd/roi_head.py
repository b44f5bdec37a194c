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
        out = torch.zeros(n, C, out_h, out_w, device=rois.device)
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
        grads = [torch.zeros(s, device=gout.device).contiguous(memory_format=fmt) for s in ctx.shapes]
        n = rois.shape[0]
        if n:
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
