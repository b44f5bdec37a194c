"""GPU tier: DeformConv2d (gfx950 sampling kernels + fp32 MFMA GEMMs) against
 (a) the reference's known-answer vectors (mmcv/tests/test_ops/test_deform_conv.py:15-36: 1x1x3x3 input, 2x2 kernel,
     fixed conv_offset weights, batch 10, im2col_step 2; rtol 1e-3 as in the reference test),
 (b) the compiled reference CPU op (oracle/_ref) on random S2ANet-like shapes when its .so travelled,
 (c) torch's plain conv2d when all offsets are zero (size-independent property).
Tolerance: fp32, 1e-4 rel forward / 1e-3 backward (atomics + f32 FMA chains of K up to 2304)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

INPUT = [[[[1., 2., 3.], [0., 1., 2.], [3., 5., 2.]]]]
OFFSET_WEIGHT = [[[0.1, 0.4, 0.6, 0.1]], [[0.3, 0.2, 0.1, 0.3]], [[0.5, 0.5, 0.2, 0.8]], [[0.8, 0.3, 0.9, 0.1]],
                 [[0.3, 0.1, 0.2, 0.5]], [[0.3, 0.7, 0.5, 0.3]], [[0.6, 0.2, 0.5, 0.3]], [[0.4, 0.1, 0.8, 0.4]]]
OFFSET_BIAS = [0.7, 0.1, 0.8, 0.5, 0.6, 0.5, 0.4, 0.7]
DEFORM_WEIGHT = [[[0.4, 0.2, 0.1, 0.9]]]
GT_OUT = [[[[1.650, 0.], [0.000, 0.]]]]
GT_X_GRAD = [[[[-0.666, 0.204, 0.000], [0.030, -0.416, 0.012], [0.000, 0.252, 0.129]]]]
GT_OFFSET_WEIGHT_GRAD = [[[[1.44, 2.88], [0.00, 1.44]]], [[[-0.72, -1.44], [0.00, -0.72]]],
                         [[[0.00, 0.00], [0.00, 0.00]]], [[[0.00, 0.00], [0.00, 0.00]]],
                         [[[-0.10, -0.20], [0.00, -0.10]]], [[[-0.08, -0.16], [0.00, -0.08]]],
                         [[[-0.54, -1.08], [0.00, -0.54]]], [[[-0.54, -1.08], [0.00, -0.54]]]]
GT_OFFSET_BIAS_GRAD = [1.44, -0.72, 0., 0., -0.10, -0.08, -0.54, -0.54]
GT_DEFORM_WEIGHT_GRAD = [[[[3.62, 0.], [0.40, 0.18]]]]


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / (b.abs().max() + 1e-12))


@pytest.mark.parametrize('im2col_step', [2, 10, 32])
def test_deform_conv_golden(im2col_step):
    from sm3det_amd.mmcv_deform_conv import DeformConv2d, DeformConv2dPack
    bs = 10
    x = torch.tensor(np.repeat(INPUT, bs, axis=0), device='cuda', dtype=torch.float32, requires_grad=True)
    model = DeformConv2dPack(in_channels=1, out_channels=1, kernel_size=2, stride=1, padding=0,
                             im2col_step=im2col_step)
    model.conv_offset.weight.data = torch.Tensor(OFFSET_WEIGHT).reshape(8, 1, 2, 2)
    model.conv_offset.bias.data = torch.Tensor(OFFSET_BIAS).reshape(8)
    model.weight.data = torch.Tensor(DEFORM_WEIGHT).reshape(1, 1, 2, 2)
    model.cuda()
    out = model(x)
    out.backward(torch.ones_like(out))
    thr = 1e-3
    assert np.allclose(out.detach().cpu().numpy(), np.repeat(GT_OUT, bs, axis=0), thr)
    assert np.allclose(x.grad.cpu().numpy(), np.repeat(GT_X_GRAD, bs, axis=0), thr)
    assert np.allclose(model.conv_offset.weight.grad.cpu().numpy() / bs, GT_OFFSET_WEIGHT_GRAD, thr)
    assert np.allclose(model.conv_offset.bias.grad.cpu().numpy() / bs, GT_OFFSET_BIAS_GRAD, thr)
    assert np.allclose(model.weight.grad.cpu().numpy() / bs, GT_DEFORM_WEIGHT_GRAD, thr)
    # constructor assertions of the reference test (:96-104)
    m = DeformConv2d(1, 1, 2, stride=1, padding=0)
    assert not hasattr(m, 'bias')
    with pytest.raises(AssertionError):
        DeformConv2d(1, 1, 2, stride=1, padding=0, bias=True)
    with pytest.raises(AssertionError):
        DeformConv2d(3, 2, 3, groups=2)
    with pytest.raises(AssertionError):
        DeformConv2d(3, 4, 3, groups=3)


@pytest.mark.parametrize('B,Cin,Cout,H,W,k,stride,pad,dil,groups,dg,step',
                         [(2, 32, 32, 20, 24, 3, 1, 1, 1, 1, 1, 2), (4, 8, 12, 13, 11, 3, 2, 1, 1, 2, 2, 2),
                          (1, 6, 4, 9, 9, 3, 1, 2, 2, 1, 3, 1)])
def test_deform_conv_vs_compiled_reference(B, Cin, Cout, H, W, k, stride, pad, dil, groups, dg, step):
    try:
        from oracle import build_ref
        ref = build_ref.load_ref()
    except Exception:
        pytest.skip('oracle/_ref .so did not travel')
    from sm3det_amd import mmcv_ext as ext
    g = torch.Generator().manual_seed(0)
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    x = torch.randn(B, Cin, H, W, generator=g)
    off = torch.randn(B, dg * 2 * k * k, Ho, Wo, generator=g) * 2
    w = torch.randn(Cout, Cin // groups, k, k, generator=g) * 0.2
    go = torch.randn(B, Cout, Ho, Wo, generator=g)
    args = (k, k, stride, stride, pad, pad, dil, dil, groups, dg)
    # reference (CPU)
    out_r = torch.zeros(B, Cout, Ho, Wo)
    ref.deform_conv_forward(x, w, off, out_r, torch.zeros(0), torch.zeros(0), *args, step)
    gi_r, goff_r, gw_r = torch.zeros_like(x), torch.zeros_like(off), torch.zeros_like(w)
    ref.deform_conv_backward_input(x, off, go.clone(), gi_r, goff_r, w, torch.zeros(0), *args, step)
    ref.deform_conv_backward_parameters(x, off, go.clone(), gw_r, torch.zeros(0), torch.zeros(0), *args, 1.0, step)
    # ours
    xc, oc, wc, goc = x.cuda(), off.cuda(), w.cuda(), go.cuda()
    out = torch.zeros(B, Cout, Ho, Wo, device='cuda')
    e = torch.zeros(0, device='cuda')
    ext.deform_conv_forward(xc, wc, oc, out, e, e, *args, step)
    gi, goff, gw = torch.zeros_like(xc), torch.zeros_like(oc), torch.zeros_like(wc)
    ext.deform_conv_backward_input(xc, oc, goc, gi, goff, wc, e, *args, step)
    ext.deform_conv_backward_parameters(xc, oc, goc, gw, e, e, *args, 1.0, step)
    assert rel(out, out_r) < 1e-4
    assert rel(gi, gi_r) < 1e-3 and rel(goff, goff_r) < 1e-3 and rel(gw, gw_r) < 1e-3


def test_zero_offsets_equal_plain_convolution():
    from sm3det_amd.mmcv_deform_conv import deform_conv2d
    x = torch.randn(2, 64, 32, 32, device='cuda', requires_grad=True)
    w = (torch.randn(64, 64, 3, 3, device='cuda') * 0.1).requires_grad_(True)
    off = torch.zeros(2, 18, 32, 32, device='cuda')
    y = deform_conv2d(x, off, w, 1, 1, 1, 1, 1, False, 32)
    go = torch.randn_like(y)
    y.backward(go)
    xr, wr = x.detach().double().requires_grad_(True), w.detach().double().requires_grad_(True)
    yr = F.conv2d(xr, wr, padding=1)
    yr.backward(go.double())
    assert rel(y, yr) < 1e-4 and rel(x.grad, xr.grad) < 1e-3 and rel(w.grad, wr.grad) < 1e-3


def test_shape_errors_raise():
    from sm3det_amd.mmcv_deform_conv import deform_conv2d
    x = torch.randn(3, 4, 8, 8, device='cuda')
    w = torch.randn(4, 4, 3, 3, device='cuda')
    with pytest.raises(AssertionError):  # batch 3 not divisible by im2col_step 2
        deform_conv2d(x, torch.zeros(3, 18, 8, 8, device='cuda'), w, 1, 1, 1, 1, 1, False, 2)
    with pytest.raises(RuntimeError):  # wrong offset channels
        deform_conv2d(x, torch.zeros(3, 10, 8, 8, device='cuda'), w, 1, 1, 1, 1, 1, False, 3)


def test_deform_conv_bench_shape_2x256x128x128_vs_compiled_reference():
    """SURVEY.md 8(d) DeformConv2d shape (the one bench.py times): x (2,256,128,128), 3x3, 256 -> 256, offsets randn*2;
    forward, input/offset gradients and weight gradient vs the reference's CPU op."""
    try:
        from oracle import build_ref
        ref = build_ref.load_ref()
    except Exception:
        pytest.skip('oracle/_ref .so did not travel')
    from sm3det_amd import mmcv_ext as ext
    g = torch.Generator().manual_seed(5)
    B, C, H, W, k = 2, 256, 128, 128, 3
    x = torch.randn(B, C, H, W, generator=g)
    off = torch.randn(B, 2 * k * k, H, W, generator=g) * 2
    w = torch.randn(C, C, k, k, generator=g) * 0.02
    go = torch.randn(B, C, H, W, generator=g)
    args = (k, k, 1, 1, 1, 1, 1, 1, 1, 1)
    out_r = torch.zeros(B, C, H, W)
    ref.deform_conv_forward(x, w, off, out_r, torch.zeros(0), torch.zeros(0), *args, 2)
    gi_r, goff_r, gw_r = torch.zeros_like(x), torch.zeros_like(off), torch.zeros_like(w)
    ref.deform_conv_backward_input(x, off, go.clone(), gi_r, goff_r, w, torch.zeros(0), *args, 2)
    ref.deform_conv_backward_parameters(x, off, go.clone(), gw_r, torch.zeros(0), torch.zeros(0), *args, 1.0, 2)
    xc, oc, wc, goc = x.cuda(), off.cuda(), w.cuda(), go.cuda()
    out = torch.zeros(B, C, H, W, device='cuda')
    e = torch.zeros(0, device='cuda')
    ext.deform_conv_forward(xc, wc, oc, out, e, e, *args, 2)
    gi, goff, gw = torch.zeros_like(xc), torch.zeros_like(oc), torch.zeros_like(wc)
    ext.deform_conv_backward_input(xc, oc, goc, gi, goff, wc, e, *args, 2)
    ext.deform_conv_backward_parameters(xc, oc, goc, gw, e, e, *args, 1.0, 2)
    assert rel(out, out_r) < 1e-4
    assert rel(gi, gi_r) < 1e-3 and rel(goff, goff_r) < 1e-3 and rel(gw, gw_r) < 1e-3


@pytest.mark.parametrize('B,Cin,Cout,H,W,k,stride,pad,dil', [(3, 48, 20, 37, 29, 3, 2, 1, 1), (2, 16, 132, 19, 23, 3, 1, 2, 2),
                                                             (1, 64, 256, 40, 40, 3, 1, 1, 1), (2, 32, 8, 9, 7, 2, 1, 0, 1)])
def test_fused_forward_equals_the_im2col_gemm_path(B, Cin, Cout, H, W, k, stride, pad, dil, monkeypatch):
    """round 4: the forward without the column matrix (csrc/deform_fused.hip: sampling inside the GEMM's operand producer)
    against the im2col + GEMM path on the same inputs -- identical column values by construction, only the fp32
    accumulation order over K differs: 2e-5 of the output scale; strides / dilations / channel counts off the tile sizes,
    sampling points far outside the image included (offsets randn * 3)."""
    from sm3det_amd import mmcv_ext as ext
    g = torch.Generator().manual_seed(B * 100 + Cin)
    Ho = (H + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    Wo = (W + 2 * pad - (dil * (k - 1) + 1)) // stride + 1
    x = torch.randn(B, Cin, H, W, generator=g).cuda()
    off = (torch.randn(B, 2 * k * k, Ho, Wo, generator=g) * 3).cuda()
    w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.2).cuda()
    args = (k, k, stride, stride, pad, pad, dil, dil, 1, 1)
    e = torch.zeros(0, device='cuda')
    outs = {}
    for mode in ('1', '0'):
        monkeypatch.setenv('SM3_DEFORM_FUSED', mode)
        out = torch.full((B, Cout, Ho, Wo), float('nan'), device='cuda')
        ext.deform_conv_forward(x, w, off, out, e, e, *args, B)
        outs[mode] = out
    assert bool(torch.isfinite(outs['1']).all())
    assert rel(outs['1'], outs['0']) < 2e-5
