"""Host-side mirror of ``mmcv.ops.{deform_conv2d, DeformConv2d, DeformConv2dPack}``
(mmcv/mmcv/ops/deform_conv.py:22-192, 195-330, 334-400) on the gfx950 kernels.  Same constructor arguments,
``forward(x, offset)`` contract, ``im2col_step`` divisibility assertion and "no bias" restriction."""
import math

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.autograd import Function
from torch.autograd.function import once_differentiable
from torch.nn.modules.utils import _pair, _single

from . import mmcv_ext as ext_module


class DeformConv2dFunction(Function):
    @staticmethod
    def forward(ctx, input, offset, weight, stride=1, padding=0, dilation=1, groups=1, deform_groups=1, bias=False,
                im2col_step=32):
        if input is not None and input.dim() != 4:
            raise ValueError(f'Expected 4D tensor as input, got {input.dim()}D tensor instead.')
        assert bias is False, 'Only support bias is False.'
        ctx.stride, ctx.padding, ctx.dilation = _pair(stride), _pair(padding), _pair(dilation)
        ctx.groups, ctx.deform_groups, ctx.im2col_step = groups, deform_groups, im2col_step
        input = input.type_as(offset).contiguous()
        weight = weight.type_as(input).contiguous()
        offset = offset.contiguous()
        ctx.save_for_backward(input, offset, weight)
        output = input.new_empty(DeformConv2dFunction._output_size(ctx, input, weight))
        ctx.bufs_ = [input.new_empty(0), input.new_empty(0)]  # columns, ones (placeholders, as in the reference)
        cur = min(ctx.im2col_step, input.size(0))
        assert (input.size(0) % cur) == 0, 'batch size must be divisible by im2col_step'
        ext_module.deform_conv_forward(
            input, weight, offset, output, ctx.bufs_[0], ctx.bufs_[1], kW=weight.size(3), kH=weight.size(2),
            dW=ctx.stride[1], dH=ctx.stride[0], padW=ctx.padding[1], padH=ctx.padding[0], dilationW=ctx.dilation[1],
            dilationH=ctx.dilation[0], group=ctx.groups, deformable_group=ctx.deform_groups, im2col_step=cur)
        return output

    @staticmethod
    @once_differentiable
    def backward(ctx, grad_output):
        input, offset, weight = ctx.saved_tensors
        grad_input = grad_offset = grad_weight = None
        cur = min(ctx.im2col_step, input.size(0))
        assert (input.size(0) % cur) == 0, 'batch size must be divisible by im2col_step'
        grad_output = grad_output.contiguous()
        kw = dict(kW=weight.size(3), kH=weight.size(2), dW=ctx.stride[1], dH=ctx.stride[0], padW=ctx.padding[1],
                  padH=ctx.padding[0], dilationW=ctx.dilation[1], dilationH=ctx.dilation[0], group=ctx.groups,
                  deformable_group=ctx.deform_groups)
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            grad_input = torch.zeros_like(input)
            grad_offset = torch.zeros_like(offset)
            ext_module.deform_conv_backward_input(input, offset, grad_output, grad_input, grad_offset, weight,
                                                  ctx.bufs_[0], im2col_step=cur, **kw)
        if ctx.needs_input_grad[2]:
            grad_weight = torch.zeros_like(weight)
            ext_module.deform_conv_backward_parameters(input, offset, grad_output, grad_weight, ctx.bufs_[0],
                                                       ctx.bufs_[1], scale=1, im2col_step=cur, **kw)
        return grad_input, grad_offset, grad_weight, None, None, None, None, None, None, None

    @staticmethod
    def _output_size(ctx, input, weight):
        channels = weight.size(0)
        output_size = (input.size(0), channels)
        for d in range(input.dim() - 2):
            in_size = input.size(d + 2)
            pad = ctx.padding[d]
            kernel = ctx.dilation[d] * (weight.size(d + 2) - 1) + 1
            output_size += ((in_size + (2 * pad) - kernel) // ctx.stride[d] + 1,)
        if not all(map(lambda s: s > 0, output_size)):
            raise ValueError('convolution input is too small (output would be ' +
                             'x'.join(map(str, output_size)) + ')')
        return output_size


deform_conv2d = DeformConv2dFunction.apply


class DeformConv2d(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 deform_groups=1, bias=False, im2col_step=32, deformable_groups=None):
        super().__init__()
        if deformable_groups is not None:  # deprecated alias (deform_conv.py:225-226)
            deform_groups = deformable_groups
        assert not bias, f'bias={bias} is not supported in DeformConv2d.'
        assert in_channels % groups == 0, f'in_channels {in_channels} cannot be divisible by groups {groups}'
        assert out_channels % groups == 0, f'out_channels {out_channels} cannot be divisible by groups {groups}'
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride = _pair(kernel_size), _pair(stride)
        self.padding, self.dilation = _pair(padding), _pair(dilation)
        self.groups, self.deform_groups, self.im2col_step = groups, deform_groups, im2col_step
        self.transposed = False
        self.output_padding = _single(0)
        self.weight = nn.Parameter(torch.Tensor(out_channels, in_channels // self.groups, *self.kernel_size))
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.kaiming_uniform_(self.weight, nonlinearity='relu')

    def forward(self, x, offset):
        input_pad = (x.size(2) < self.kernel_size[0]) or (x.size(3) < self.kernel_size[1])
        if input_pad:
            pad_h = max(self.kernel_size[0] - x.size(2), 0)
            pad_w = max(self.kernel_size[1] - x.size(3), 0)
            x = F.pad(x, (0, pad_w, 0, pad_h), 'constant', 0).contiguous()
            offset = F.pad(offset, (0, pad_w, 0, pad_h), 'constant', 0).contiguous()
        out = deform_conv2d(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                            self.deform_groups, False, self.im2col_step)
        if input_pad:
            out = out[:, :, :out.size(2) - pad_h, :out.size(3) - pad_w].contiguous()
        return out

    def __repr__(self):
        return (f'{self.__class__.__name__}(in_channels={self.in_channels},\nout_channels={self.out_channels},\n'
                f'kernel_size={self.kernel_size},\nstride={self.stride},\npadding={self.padding},\n'
                f'dilation={self.dilation},\ngroups={self.groups},\ndeform_groups={self.deform_groups},\nbias=False)')


class DeformConv2dPack(DeformConv2d):
    """DeformConv2d with its own offset-predicting conv (zero-initialised), deform_conv.py:334-400."""
    _version = 2

    def __init__(self, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.conv_offset = nn.Conv2d(self.in_channels, self.deform_groups * 2 * self.kernel_size[0] *
                                     self.kernel_size[1], kernel_size=self.kernel_size, stride=_pair(self.stride),
                                     padding=_pair(self.padding), dilation=_pair(self.dilation), bias=True)
        self.init_offset()

    def init_offset(self):
        self.conv_offset.weight.data.zero_()
        self.conv_offset.bias.data.zero_()

    def forward(self, x):
        offset = self.conv_offset(x)
        return deform_conv2d(x, offset, self.weight, self.stride, self.padding, self.dilation, self.groups,
                             self.deform_groups, False, self.im2col_step)
