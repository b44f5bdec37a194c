"""``mmcv.ops.{deform_conv2d, DeformConv2d, DeformConv2dPack}`` on the gfx950 kernels.

Call surface of the reference front-end (mmcv/mmcv/ops/deform_conv.py: function :22-192, module :195-330, pack :334-400):
same constructor / call arguments, the ``im2col_step`` divisibility rule, bias-free weights, and exactly the three native
calls the reference issues per forward / backward (``tests/test_ref_wrappers.py`` compares them one for one with the
reference's own wrapper).  The bodies are this package's: one geometry record shared by forward and backward, and the
offset-predicting convolution of ``DeformConv2dPack`` runs as im2col + this library's GEMM (no MIOpen convolution).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair

from . import mmcv_ext as ext_module


def _geometry(weight, stride, padding, dilation, groups, deform_groups):
    """the keyword block every native entry point takes (pybind.cpp:38-57: W before H)"""
    (sh, sw), (ph, pw), (dh, dw) = _pair(stride), _pair(padding), _pair(dilation)
    return dict(kW=weight.size(3), kH=weight.size(2), dW=sw, dH=sh, padW=pw, padH=ph, dilationW=dw, dilationH=dh,
                group=groups, deformable_group=deform_groups)


def _out_hw(size, k, stride, pad, dil):
    return (size + 2 * pad - (dil * (k - 1) + 1)) // stride + 1


class _DeformConv(Function):
    @staticmethod
    def forward(ctx, x, offset, weight, geo, im2col_step):
        if x is not None and x.dim() != 4:
            raise ValueError(f'Expected 4D tensor as input, got {x.dim()}D tensor instead.')
        x = x.type_as(offset).contiguous()
        weight = weight.type_as(x).contiguous()
        offset = offset.contiguous()
        step = min(im2col_step, x.size(0))
        assert x.size(0) % step == 0, 'batch size must be divisible by im2col_step'
        ho = _out_hw(x.size(2), geo['kH'], geo['dH'], geo['padH'], geo['dilationH'])
        wo = _out_hw(x.size(3), geo['kW'], geo['dW'], geo['padW'], geo['dilationW'])
        if min(x.size(0), weight.size(0), ho, wo) <= 0:
            raise ValueError(f'convolution input is too small (output would be {x.size(0)}x{weight.size(0)}x{ho}x{wo})')
        out = x.new_empty(x.size(0), weight.size(0), ho, wo)
        # the two scratch tensors of the native interface (`columns`, `ones`): empty placeholders, the callee sizes its own
        ctx.scratch = (x.new_empty(0), x.new_empty(0))
        ctx.geo, ctx.step = geo, step
        ctx.save_for_backward(x, offset, weight)
        ext_module.deform_conv_forward(x, weight, offset, out, ctx.scratch[0], ctx.scratch[1], im2col_step=step, **geo)
        return out

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_out):
        x, offset, weight = ctx.saved_tensors
        grad_out = grad_out.contiguous()
        gx = goff = gw = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            gx, goff = torch.zeros_like(x), torch.zeros_like(offset)
            ext_module.deform_conv_backward_input(x, offset, grad_out, gx, goff, weight, ctx.scratch[0],
                                                  im2col_step=ctx.step, **ctx.geo)
        if ctx.needs_input_grad[2]:
            gw = torch.zeros_like(weight)
            ext_module.deform_conv_backward_parameters(x, offset, grad_out, gw, ctx.scratch[0], ctx.scratch[1], scale=1,
                                                       im2col_step=ctx.step, **ctx.geo)
        return gx, goff, gw, None, None


def deform_conv2d(input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deform_groups=1, bias=False,
                  im2col_step=32):
    """Deformable convolution v1: ``offset`` is ``(B, deform_groups * 2 * kH * kW, Ho, Wo)``; no bias."""
    assert bias is False, 'Only support bias is False.'
    return _DeformConv.apply(input, offset, weight, _geometry(weight, stride, padding, dilation, groups, deform_groups),
                             im2col_step)


class DeformConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deform_groups=1, bias=False, im2col_step=32, deformable_groups=None):
        super().__init__()
        assert not bias, f'bias={bias} is not supported in DeformConv2d.'
        assert in_channels % groups == 0, f'in_channels {in_channels} cannot be divisible by groups {groups}'
        assert out_channels % groups == 0, f'out_channels {out_channels} cannot be divisible by groups {groups}'
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.groups, self.im2col_step = groups, im2col_step
        self.deform_groups = deform_groups if deformable_groups is None else deformable_groups  # deprecated alias
        self.transposed, self.output_padding = False, (0,)
        self.weight = nn.Parameter(torch.empty(out_channels, in_channels // groups, *self.kernel_size))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, nonlinearity='relu')

    def _apply_op(self, x, offset):
        return deform_conv2d(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                             self.deform_groups, False, self.im2col_step)

    def forward(self, x, offset):
        # a map smaller than the kernel is zero-extended on its bottom / right edge (offsets too) and the result cropped back
        grow_h = max(self.kernel_size[0] - x.size(2), 0)
        grow_w = max(self.kernel_size[1] - x.size(3), 0)
        if not (grow_h or grow_w):
            return self._apply_op(x, offset)
        x = F.pad(x, (0, grow_w, 0, grow_h)).contiguous()
        offset = F.pad(offset, (0, grow_w, 0, grow_h)).contiguous()
        out = self._apply_op(x, offset)
        return out[:, :, :out.size(2) - grow_h, :out.size(3) - grow_w].contiguous()

    def extra_repr(self):
        return (f'in_channels={self.in_channels}, out_channels={self.out_channels}, kernel_size={self.kernel_size}, '
                f'stride={self.stride}, padding={self.padding}, dilation={self.dilation}, groups={self.groups}, '
                f'deform_groups={self.deform_groups}, bias=False')


class _OffsetConv(nn.Module):
    """holder of ``conv_offset.weight (Cout,Cin,kH,kW)`` / ``conv_offset.bias`` (the reference's key names and shapes, zero
    initialised: deform_conv.py:371-373); applied as im2col + the library's GEMM"""

    def __init__(self, cin, cout, kernel_size, stride, padding, dilation):
        super().__init__()
        self.geo = dict(kernel_size=kernel_size, dilation=dilation, padding=padding, stride=stride)
        self.weight = nn.Parameter(torch.zeros(cout, cin, *kernel_size))
        self.bias = nn.Parameter(torch.zeros(cout))

    def forward(self, x):
        from .backbone_ops import linear
        B, _, H, W = x.shape
        (kh, kw), (sh, sw) = self.geo['kernel_size'], self.geo['stride']
        ho = _out_hw(H, kh, sh, self.geo['padding'][0], self.geo['dilation'][0])
        wo = _out_hw(W, kw, sw, self.geo['padding'][1], self.geo['dilation'][1])
        cols = F.unfold(x, **self.geo).transpose(1, 2).reshape(B * ho * wo, -1)  # (positions, Cin*kH*kW)
        K, N = cols.size(1), self.weight.size(0)
        k32, n32 = (K + 31) // 32 * 32, (N + 31) // 32 * 32                      # the GEMM's K / N granules
        y = linear(F.pad(cols, (0, k32 - K)).contiguous(), F.pad(self.weight.reshape(N, K), (0, k32 - K, 0, n32 - N)),
                   F.pad(self.bias, (0, n32 - N)))
        return y[:, :N].reshape(B, ho, wo, N).permute(0, 3, 1, 2).contiguous()


class DeformConv2dPack(DeformConv2d):
    """DeformConv2d that predicts its own offsets with a zero-initialised convolution of the same geometry."""
    _version = 2

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.conv_offset = _OffsetConv(self.in_channels, self.deform_groups * 2 * self.kernel_size[0] * self.kernel_size[1],
                                       self.kernel_size, self.stride, self.padding, self.dilation)

    def init_offset(self):
        with torch.no_grad():
            self.conv_offset.weight.zero_()
            self.conv_offset.bias.zero_()

    def forward(self, x):
        return self._apply_op(x, self.conv_offset(x))
