"""``MultitaskFPN`` on the MI355X kernels -- SURVEY.md 8(f) row 2: the step right after the backbone, in the layout
the backbone emits (NHWC tokens), so no layout change happens between them.

Mirror of ``mmrotate/models/necks/Multitask_FPN.py`` (class :15, ``__init__`` :17-106, ``forward`` :108-162): same
constructor kwargs, ``forward(inputs, start_level=None, add_extra_convs=None)`` (the detector calls it per modality
with different ``start_level``: ``trisource_*_detector.py:151-167``), same ``state_dict`` keys and shapes
(``lateral_convs.{i}.conv.{weight,bias}``, ``fpn_convs.{i}.conv.{weight,bias}``).  Every SM3Det config builds it with
``conv_cfg = norm_cfg = act_cfg = None``, i.e. each ``ConvModule`` is ``nn.Conv2d`` + bias; anything else raises.

Execution: laterals = NT GEMM + bias on the fp32 matrix cores (``backbone_ops.linear``); top-down merge = one HBM pass
(``sm3_upsample2x_add``); 3x3 / stride-1|2 output convs = implicit GEMMs (``sm3_conv3x3_nhwc_*``: the im2col operand is
gathered tap by tap while the tile is loaded).  Backward is hand-written per op (dgrad = transposed-conv gather, wgrad =
split-K TN with the gathered operand, bias grad = column sums).  Weights are stored in the kernels' layout --
``(Cout, 3, 3, Cin)`` / ``(Cout, Cin)`` -- and converted in ``state_dict()`` / ``load_state_dict()``.

Inputs: the backbone's outputs (logically NCHW over NHWC memory = ``torch.channels_last``) or any NCHW tensor (copied
to NHWC once).  Outputs: logically NCHW over NHWC memory.  No CPU fallback.
"""
import math

import torch
import torch.nn as nn
from torch.autograd import Function

from . import _lib
from . import _lib_backbone as LB
from . import backbone_ops as ops
from . import level_streams
from .registry import ROTATED_NECKS as _REG

call, colsum = LB.call, LB.colsum


def _e(*shape, like):
    return torch.empty(*shape, device=like.device, dtype=torch.float32)


# ------------------------------------------------------------------------------------------------ functions
class _Conv3x3(Function):
    """y (B,Ho,Wo,Cout) = conv3x3(x (B,H,W,Cin), w (Cout,3,3,Cin), padding 1, stride s) + bias"""

    @staticmethod
    def forward(ctx, x, w, b, stride, relu=False):
        _lib.require_gpu(x, w, b)
        x, w = x.contiguous(), w.contiguous()
        b = b.contiguous() if b is not None else None  # bias-free: ConvModule in front of a norm layer
        B, H, W, Cin = x.shape
        Cout = w.shape[0]
        Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
        y = _e(B, Ho, Wo, Cout, like=x)
        nb = _lib.lib().sm3_conv3x3_nhwc_workspace_bytes(B, H, W, Cin, Cout, stride, 0)
        ws = _lib.workspace(nb, x.device)
        call('conv3x3_nhwc_fwd', x, w, b, y, B, H, W, Cin, Cout, stride, int(relu), ws, nb,
             flops=2.0 * B * Ho * Wo * Cout * 9 * Cin)
        ctx.save_for_backward(x, w, y if relu else None)
        ctx.stride, ctx.relu, ctx.has_bias = stride, relu, b is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, y = ctx.saved_tensors
        s = ctx.stride
        dy = dy.contiguous()
        if ctx.relu:  # gradient of max(., 0): pass where the saved output is positive
            dpre = _e(*dy.shape, like=dy)
            call('relu_bwd', dy, y, dpre, dy.numel())
            dy = dpre
        B, H, W, Cin = x.shape
        Cout = w.shape[0]
        Ho, Wo = dy.shape[1], dy.shape[2]
        fl = 2.0 * B * Ho * Wo * Cout * 9 * Cin
        dx = None
        if ctx.needs_input_grad[0]:
            dx = _e(B, H, W, Cin, like=x)
            nb = _lib.lib().sm3_conv3x3_nhwc_workspace_bytes(B, H, W, Cin, Cout, s, 1)
            ws = _lib.workspace(nb, x.device)
            call('conv3x3_nhwc_bwd_input', dy, w, dx, B, H, W, Cin, Cout, s, ws, nb, flops=fl)
        dw = _e(Cout, 3, 3, Cin, like=x)
        nb = _lib.lib().sm3_conv3x3_nhwc_bwd_weight_workspace_bytes(B, H, W, Cin, Cout, s)
        ws = _lib.workspace(nb, x.device)
        call('conv3x3_nhwc_bwd_weight', x, dy, dw, B, H, W, Cin, Cout, s, ws, nb, flops=fl)
        db = None
        if ctx.has_bias:
            db = _e(Cout, like=x)
            colsum(dy.view(-1, Cout), B * Ho * Wo, Cout, db)
        return dx, dw, db, None, None


class _UpsampleAdd(Function):
    """fine (B,H,W,C) + nearest-2x(coarse (B,H/2,W/2,C))"""

    @staticmethod
    def forward(ctx, fine, coarse):
        _lib.require_gpu(fine, coarse)
        fine, coarse = fine.contiguous(), coarse.contiguous()
        B, H, W, C = fine.shape
        if coarse.shape != (B, H // 2, W // 2, C) or H % 2 or W % 2:
            raise _lib.SM3Error(f'top-down merge needs an exact 2x pyramid: fine {tuple(fine.shape)}, coarse '
                                f'{tuple(coarse.shape)}')
        out = _e(B, H, W, C, like=fine)
        call('upsample2x_add', fine, coarse, out, B, H, W, C)
        return out

    @staticmethod
    def backward(ctx, dout):
        dout = dout.contiguous()
        B, H, W, C = dout.shape
        dcoarse = None
        if ctx.needs_input_grad[1]:
            dcoarse = _e(B, H // 2, W // 2, C, like=dout)
            call('sumpool2x_add', dout, None, dcoarse, B, H // 2, W // 2, C)
        return dout, dcoarse


def conv3x3_nhwc(x, w, b, stride=1, relu=False):
    return _Conv3x3.apply(x, w, b, stride, relu)


def upsample2x_add(fine, coarse):
    return _UpsampleAdd.apply(fine, coarse)


# ------------------------------------------------------------------------------------------------ modules
class _Conv(nn.Module):
    """Parameter holder standing where the reference has ``ConvModule.conv`` (an ``nn.Conv2d``): weight kept in the
    kernel layout, ``(Cout, k, k, Cin)`` (k = 3) or ``(Cout, Cin)`` (k = 1); ``state_dict`` uses ``(Cout, Cin, k, k)``."""

    def __init__(self, cin, cout, k, stride=1):
        super().__init__()
        self.in_channels, self.out_channels, self.k, self.stride = cin, cout, k, stride
        ref = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2)  # same default init / RNG consumption
        self.weight = nn.Parameter(self._to_kernel(ref.weight.data))
        self.bias = nn.Parameter(ref.bias.data.clone())

    def _to_kernel(self, w4):
        w = w4.permute(0, 2, 3, 1).contiguous()
        return w.reshape(self.out_channels, self.in_channels) if self.k == 1 else w

    def _to_reference(self, w):
        return w.reshape(self.out_channels, self.k, self.k, self.in_channels).permute(0, 3, 1, 2).contiguous()

    def _save_to_state_dict(self, destination, prefix, keep_vars):
        super()._save_to_state_dict(destination, prefix, keep_vars)
        destination[prefix + 'weight'] = self._to_reference(destination[prefix + 'weight'])

    def _load_from_state_dict(self, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                              error_msgs):
        key = prefix + 'weight'
        if key in state_dict and state_dict[key].dim() == 4 and state_dict[key].shape[1] == self.in_channels:
            state_dict[key] = self._to_kernel(state_dict[key])
        super()._load_from_state_dict(state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys,
                                      error_msgs)

    def xavier_uniform_(self):
        """Xavier/uniform with the fans of the REFERENCE shape (Cout, Cin, k, k), bias 0 (mmcv xavier_init)."""
        fan_in, fan_out = self.in_channels * self.k * self.k, self.out_channels * self.k * self.k
        a = math.sqrt(3.0) * math.sqrt(2.0 / float(fan_in + fan_out))
        with torch.no_grad():
            self.weight.uniform_(-a, a)
            self.bias.zero_()


class ConvModule(nn.Module):
    """``mmcv.cnn.ConvModule`` restricted to conv + bias (norm_cfg = act_cfg = None), keeping the ``.conv`` child so the
    parameter names match."""

    def __init__(self, cin, cout, k, stride=1, padding=0, conv_cfg=None, norm_cfg=None, act_cfg=None, inplace=False):
        super().__init__()
        if conv_cfg is not None or norm_cfg is not None or act_cfg is not None:
            raise NotImplementedError('MultitaskFPN on MI355X: conv_cfg / norm_cfg / act_cfg must be None (every '
                                      'SM3Det config)')
        if k not in (1, 3) or padding != k // 2 or (k == 1 and stride != 1):
            raise NotImplementedError('only 1x1 and padded 3x3 (stride 1 or 2) convolutions are implemented')
        self.conv = _Conv(cin, cout, k, stride)

    def forward_nhwc(self, x):
        c = self.conv
        if c.k == 1:
            B, H, W, Cin = x.shape
            return ops.linear(x.reshape(-1, Cin), c.weight, c.bias).view(B, H, W, c.out_channels)
        return conv3x3_nhwc(x, c.weight, c.bias, c.stride)

    def forward(self, x):
        return _to_nchw_view(self.forward_nhwc(_to_nhwc(x)))


def _to_nhwc(x):
    """logically-NCHW tensor -> (B,H,W,C) contiguous (free for channels_last inputs)"""
    return x.permute(0, 2, 3, 1).contiguous()


def _to_nchw_view(x):
    return x.permute(0, 3, 1, 2)


@_REG.register_module()
class MultitaskFPN(nn.Module):
    def __init__(self, in_channels, out_channels, num_outs, start_level=0, end_level=-1, extra_level=0,
                 add_extra_convs=False, relu_before_extra_convs=False, no_norm_on_lateral=False, conv_cfg=None,
                 norm_cfg=None, act_cfg=None, upsample_cfg=dict(mode='nearest'),
                 init_cfg=dict(type='Xavier', layer='Conv2d', distribution='uniform')):
        super().__init__()
        assert isinstance(in_channels, list)
        self.init_cfg = init_cfg
        self.in_channels = in_channels
        self.out_channels = out_channels
        self.num_ins = len(in_channels)
        self.num_outs = num_outs
        self.relu_before_extra_convs = relu_before_extra_convs
        self.no_norm_on_lateral = no_norm_on_lateral
        self.fp16_enabled = False
        self.upsample_cfg = dict(upsample_cfg)
        if self.upsample_cfg.get('mode', 'nearest') != 'nearest' or 'scale_factor' in self.upsample_cfg:
            raise NotImplementedError("only upsample_cfg=dict(mode='nearest') (every SM3Det config) is implemented")
        if relu_before_extra_convs:
            raise NotImplementedError('relu_before_extra_convs=True is not used by any SM3Det config')
        if end_level == -1 or end_level == self.num_ins - 1:  # reference :44-52
            self.backbone_end_level = self.num_ins
            assert num_outs >= self.num_ins - start_level
        else:
            self.backbone_end_level = end_level + 1
            assert end_level < self.num_ins
            assert num_outs == end_level - start_level + 1
        self.start_level = start_level
        self.end_level = end_level
        self.extra_level = extra_level
        self.add_extra_convs = add_extra_convs
        assert isinstance(add_extra_convs, (str, bool))
        if isinstance(add_extra_convs, str):
            assert add_extra_convs in ('on_input', 'on_lateral', 'on_output')
        elif add_extra_convs:
            self.add_extra_convs = 'on_input'
        self.lateral_convs = nn.ModuleList()
        self.fpn_convs = nn.ModuleList()
        for i in range(self.start_level, self.backbone_end_level):  # reference :66-86
            self.lateral_convs.append(ConvModule(in_channels[i], out_channels, 1, conv_cfg=conv_cfg,
                                                 norm_cfg=norm_cfg if not no_norm_on_lateral else None,
                                                 act_cfg=act_cfg))
            self.fpn_convs.append(ConvModule(out_channels, out_channels, 3, padding=1, conv_cfg=conv_cfg,
                                             norm_cfg=norm_cfg, act_cfg=act_cfg))
        extra_levels = num_outs - self.backbone_end_level + self.extra_level  # reference :89-106
        if self.add_extra_convs and extra_levels >= 1:
            for i in range(extra_levels):
                if i == 0 and self.add_extra_convs == 'on_input':
                    cin = self.in_channels[self.backbone_end_level - 1]
                else:
                    cin = out_channels
                self.fpn_convs.append(ConvModule(cin, out_channels, 3, stride=2, padding=1, conv_cfg=conv_cfg,
                                                 norm_cfg=norm_cfg, act_cfg=act_cfg))

    def init_weights(self):
        """init_cfg = Xavier / uniform on every Conv2d (reference :31-32)."""
        for m in self.modules():
            if isinstance(m, _Conv):
                m.xavier_uniform_()

    def forward(self, inputs, start_level=None, add_extra_convs=None):
        """reference :108-162, same control flow; tensors are (B,H,W,C) inside."""
        if start_level is None:
            start_level = self.start_level
        if add_extra_convs is None:
            add_extra_convs = self.add_extra_convs
        xs = [_to_nhwc(t) for t in inputs]
        lcs = list(self.lateral_convs)[start_level:]
        laterals = level_streams.map_levels(lambda x, lc: lc.forward_nhwc(x), xs[start_level:start_level + len(lcs)], lcs)
        used = len(laterals)
        for i in range(used - 1, 0, -1):  # top-down path :123-135
            laterals[i - 1] = upsample2x_add(laterals[i - 1], laterals[i])
        outs = level_streams.map_levels(lambda x, fc: fc.forward_nhwc(x), laterals,
                                        list(self.fpn_convs)[start_level:start_level + used])
        if self.num_outs > len(outs):  # :142-161
            if not add_extra_convs:
                for _ in range(self.num_outs - used):
                    outs.append(outs[-1][:, ::2, ::2, :].contiguous())  # F.max_pool2d(x, 1, stride=2)
            else:
                if add_extra_convs == 'on_input':
                    src = xs[self.backbone_end_level - 1]
                elif add_extra_convs == 'on_lateral':
                    src = laterals[-1]
                elif add_extra_convs == 'on_output':
                    src = outs[-1]
                else:
                    raise NotImplementedError
                outs.append(self.fpn_convs[used + start_level].forward_nhwc(src))
                for i in range(used + 1, self.num_outs):
                    outs.append(self.fpn_convs[i + start_level].forward_nhwc(outs[-1]))
        return tuple(_to_nchw_view(o) for o in outs)
