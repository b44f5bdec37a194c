"""DeformConv2d on MI355X: host orchestration of the three ``mmcv._ext`` entry points

    deform_conv_forward / deform_conv_backward_input / deform_conv_backward_parameters
    (pybind.cpp:38-57,501-522; reference host code mmcv/mmcv/ops/csrc/pytorch/deform_conv.cpp:140-517)

around the gfx950 sampling kernels (``sm3_deform_im2col / col2im / col2im_coord``, csrc/deform_conv.hip) and the fp32
MFMA GEMM family (the reference's per-group ``addmm_`` calls).  Same argument lists and in-place output conventions as
the reference; shape errors raise ``RuntimeError`` like its ``TORCH_CHECK``s (deform_conv.cpp:45-138).

The column matrix keeps the reference layout ``(Cin*kH*kW, im2col_step*Ho*Wo)`` but with its leading dimension rounded
up to 32 floats so that it feeds the GEMM kernels without a copy; the tiny re-layouts of weights / outputs between the
(step, C, H, W) and (C, step*H*W) views are torch glue on O(output) tensors, exactly where the reference does
``transpose_`` + ``copy_``.
"""
import os

import torch
import torch.nn.functional as F

from . import _lib
from . import _lib_backbone as LB
from ._lib import SM3Error, require_gpu


def _rup(v, m):
    return (v + m - 1) // m * m


def _shape_check(input, offset, grad_output, weight, kH, kW, dH, dW, padH, padW, dilationH, dilationW, group,
                 deformable_group):
    if weight.dim() != 4:
        raise SM3Error(f'4D weight tensor (nOutputPlane,nInputPlane,kH,kW) expected, but got: {weight.dim()}')
    if not weight.is_contiguous():
        raise SM3Error('weight tensor has to be contiguous')
    if kW <= 0 or kH <= 0:
        raise SM3Error(f'kernel size should be greater than zero, but got kH: {kH} kW: {kW}')
    if weight.size(2) != kH or weight.size(3) != kW:
        raise SM3Error('kernel size should be consistent with weight')
    if dW <= 0 or dH <= 0:
        raise SM3Error(f'stride should be greater than zero, but got dH: {dH} dW: {dW}')
    if dilationW <= 0 or dilationH <= 0:
        raise SM3Error('dilation should be greater than 0')
    if input.dim() != 4:
        raise SM3Error(f'4D input tensor expected but got: {input.dim()}')
    nIn = weight.size(1) * group
    H, W = input.size(2), input.size(3)
    nOut = weight.size(0)
    Ho = (H + 2 * padH - (dilationH * (kH - 1) + 1)) // dH + 1
    Wo = (W + 2 * padW - (dilationW * (kW - 1) + 1)) // dW + 1
    if nIn % deformable_group != 0:
        raise SM3Error('input channels must divide deformable group size')
    if Wo < 1 or Ho < 1:
        raise SM3Error(f'Given input size: ({nIn} x {H} x {W}). Calculated output size: ({nOut} x {Ho} x {Wo}). '
                       'Output size is too small')
    if input.size(1) != nIn:
        raise SM3Error(f'invalid number of input planes, expected: {nIn}, but got: {input.size(1)}')
    if H < kH or W < kW:
        raise SM3Error('input image is smaller than kernel')
    if offset.size(2) != Ho or offset.size(3) != Wo:
        raise SM3Error(f'invalid spatial size of offset, expected height: {Ho} width: {Wo}, but got height: '
                       f'{offset.size(2)} width: {offset.size(3)}')
    if offset.size(1) != deformable_group * 2 * kH * kW:
        raise SM3Error('invalid number of channels of offset')
    if offset.size(0) != input.size(0):
        raise SM3Error('invalid batch size of offset')
    if grad_output is not None:
        if grad_output.size(1) != nOut:
            raise SM3Error(f'invalid number of gradOutput planes, expected: {nOut}, but got: {grad_output.size(1)}')
        if grad_output.size(2) != Ho or grad_output.size(3) != Wo:
            raise SM3Error('invalid size of gradOutput')
    return nIn, nOut, H, W, Ho, Wo


def _f32c(*ts):
    for t in ts:
        if t.dtype != torch.float32:
            raise SM3Error('DeformConv2d kernels are float32')
        if not t.is_contiguous():
            raise SM3Error('DeformConv2d tensors must be contiguous')


def _im2col(inp_e, off_e, col, geom, ld):
    LB.call('deform_im2col', inp_e, off_e, col, *geom, ld)


def _group_view(col, group, Kg, Kp, ld):
    """(Cin*kh*kw, ld) -> per-group matrices with K padded to the GEMM granule (copy only for odd K)."""
    v = col.view(group, Kg, ld)
    return v if Kp == Kg else F.pad(v, (0, 0, 0, Kp - Kg)).contiguous()


def _weight_view(weight, group, Kg, Kp):
    w = weight.reshape(group, weight.size(0) // group, Kg)
    return w.contiguous() if Kp == Kg else F.pad(w, (0, Kp - Kg)).contiguous()


def deform_conv_forward(input, weight, offset, output, columns, ones, kW, kH, dW, dH, padW, padH, dilationW,
                        dilationH, group, deformable_group, im2col_step):
    require_gpu(input, weight, offset, output)
    _f32c(input, weight, offset, output)
    nIn, nOut, H, W, Ho, Wo = _shape_check(input, offset, None, weight, kH, kW, dH, dW, padH, padW, dilationH,
                                           dilationW, group, deformable_group)
    B = input.size(0)
    step = im2col_step
    if B % step:
        raise SM3Error('batch size must be divisible by im2col_step')
    L = _lib.lib()
    if (os.environ.get('SM3_DEFORM_FUSED', '1') != '0' and
            L.sm3_deform_conv_fwd_fused_supported(nIn, nOut, kH, kW, group, deformable_group)):
        # no column matrix: the sampling runs inside the MFMA GEMM's operand producer (csrc/deform_fused.hip).  The result
        # does not depend on im2col_step (the reference only chunks the batch with it).
        with torch.cuda.device(input.device):
            x_nhwc = torch.empty(B, H, W, nIn, device=input.device)
            _lib.check(L.sm3_transpose_f32(input.data_ptr(), x_nhwc.data_ptr(), B, nIn, H * W, _lib.stream_ptr()),
                       'transpose_f32')
            w_t = weight.permute(2, 3, 1, 0).reshape(kH * kW * nIn, nOut).contiguous()
            LB.call('deform_conv_fwd_fused', x_nhwc, offset, w_t, output, B, nIn, H, W, nOut, kH, kW, padH, padW, dH, dW,
                    dilationH, dilationW, flops=2.0 * B * Ho * Wo * nOut * nIn * kH * kW)
        return
    ncols = step * Ho * Wo
    ld = _rup(ncols, 32)
    Kg = nIn // group * kH * kW
    Kp = _rup(Kg, 32)
    geom = (nIn, H, W, kH, kW, padH, padW, dH, dW, dilationH, dilationW, step, deformable_group)
    col = torch.zeros(nIn * kH * kW, ld, device=input.device)
    wg = _weight_view(weight, group, Kg, Kp)
    Mg = nOut // group
    with torch.cuda.device(input.device):
        for elt in range(B // step):
            sl = slice(elt * step, (elt + 1) * step)
            _im2col(input[sl], offset[sl], col, geom, ld)
            cg = _group_view(col, group, Kg, Kp, ld)
            ob = torch.empty(group, Mg, ld, device=input.device)
            for g in range(group):
                LB.gemm(LB.NN, wg[g], cg[g], ob[g], Mg, ld, Kp, ldb=ld)
            # (Cout, step, Ho, Wo) -> (step, Cout, Ho, Wo): the reference's transpose_(1,2) + copy_ (:245-247)
            output[sl].copy_(ob.view(nOut, ld)[:, :ncols].view(nOut, step, Ho, Wo).transpose(0, 1))


def _go_matrix(grad_output, sl, nOut, step, Ho, Wo, ld):
    ncols = step * Ho * Wo
    go = grad_output.new_zeros(nOut, ld)
    go[:, :ncols].view(nOut, step, Ho, Wo).copy_(grad_output[sl].transpose(0, 1))
    return go


def deform_conv_backward_input(input, offset, gradOutput, gradInput, gradOffset, weight, columns, kW, kH, dW, dH, padW,
                               padH, dilationW, dilationH, group, deformable_group, im2col_step):
    require_gpu(input, offset, gradOutput, gradInput, gradOffset, weight)
    _f32c(input, offset, gradOutput, gradInput, gradOffset, weight)
    nIn, nOut, H, W, Ho, Wo = _shape_check(input, offset, gradOutput, weight, kH, kW, dH, dW, padH, padW, dilationH,
                                           dilationW, group, deformable_group)
    B = input.size(0)
    step = im2col_step
    if B % step:
        raise SM3Error('batch size must be divisible by im2col_step')
    ld = _rup(step * Ho * Wo, 32)
    Kg = nIn // group * kH * kW
    Kp = _rup(Kg, 32)
    geom = (nIn, H, W, kH, kW, padH, padW, dH, dW, dilationH, dilationW, step, deformable_group)
    wg = _weight_view(weight, group, Kg, Kp)
    Mg = nOut // group
    with torch.cuda.device(input.device):
        for elt in range(B // step):
            sl = slice(elt * step, (elt + 1) * step)
            go = _go_matrix(gradOutput, sl, nOut, step, Ho, Wo, ld).view(group, Mg, ld)
            colp = torch.empty(group, Kp, ld, device=input.device)
            for g in range(group):  # columns[g] = W[g]^T @ gradOutput[g]  (:339-342)
                LB.gemm(LB.TN, wg[g], go[g], colp[g], Kp, ld, Mg, lda=Kp, ldb=ld, splits=1)
            col = colp if Kp == Kg else colp[:, :Kg].contiguous()
            col = col.view(nIn * kH * kW, ld)
            # both gradients in one pass over the columns, NHWC on both sides: the input is transposed once, the input
            # gradient is scattered on an NHWC scratch map (coalesced channel-vector atomics) and added into gradInput
            inp = input[sl]
            im_nhwc = torch.empty(step, H, W, nIn, device=input.device)
            _lib.check(_lib.lib().sm3_transpose_f32(inp.data_ptr(), im_nhwc.data_ptr(), step, nIn, H * W,
                                                    _lib.stream_ptr()), 'transpose_f32')
            scratch = torch.zeros(step, H, W, nIn, device=input.device)
            LB.call('deform_bwd_input_fused', col, im_nhwc, offset[sl], scratch, gradOffset[sl], *geom, ld)
            gi = gradInput[sl]
            _lib.check(_lib.lib().sm3_transpose_add_f32(scratch.data_ptr(), gi.data_ptr(), step, H * W, nIn,
                                                        _lib.stream_ptr()), 'transpose_add_f32')


def deform_conv_backward_parameters(input, offset, gradOutput, gradWeight, columns, ones, kW, kH, dW, dH, padW, padH,
                                    dilationW, dilationH, group, deformable_group, scale, im2col_step):
    require_gpu(input, offset, gradOutput, gradWeight)
    _f32c(input, offset, gradOutput, gradWeight)
    nIn, nOut, H, W, Ho, Wo = _shape_check(input, offset, gradOutput, gradWeight, kH, kW, dH, dW, padH, padW,
                                           dilationH, dilationW, group, deformable_group)
    B = input.size(0)
    step = im2col_step
    if B % step:
        raise SM3Error('batch size must be divisible by im2col_step')
    ld = _rup(step * Ho * Wo, 32)
    Kg = nIn // group * kH * kW
    Kp = _rup(Kg, 32)
    geom = (nIn, H, W, kH, kW, padH, padW, dH, dW, dilationH, dilationW, step, deformable_group)
    Mg = nOut // group
    col = torch.zeros(nIn * kH * kW, ld, device=input.device)
    gw = gradWeight.view(group, Mg, Kg)
    with torch.cuda.device(input.device):
        for elt in range(B // step):
            sl = slice(elt * step, (elt + 1) * step)
            _im2col(input[sl], offset[sl], col, geom, ld)
            cg = _group_view(col, group, Kg, Kp, ld)
            go = _go_matrix(gradOutput, sl, nOut, step, Ho, Wo, ld).view(group, Mg, ld)
            part = torch.empty(group, Mg, Kp, device=input.device)
            for g in range(group):  # gradWeight[g] += scale * gradOutput[g] @ columns[g]^T  (:478-484)
                LB.gemm(LB.NT, go[g], cg[g], part[g], Mg, Kp, ld)
            gw.add_(part[:, :, :Kg], alpha=float(scale))
