"""GPU tier: the implicit-GEMM 3x3 convolution kernels, the top-down merge kernels and the MultitaskFPN module against
torch-CPU references / the CPU oracle / the reference fixtures.  Tolerances: fp32 contractions with K up to 6912 ->
1e-4 relative (max-norm) forward, 1e-3 backward (split-K + different summation order)."""
import pytest
import torch
import torch.nn.functional as F

from tests.fpn_common import CASES, MG, check_run, load, rel_err

pytestmark = pytest.mark.gpu
FWD_TOL, BWD_TOL = 1e-4, 1e-3


@pytest.mark.parametrize('B,H,W,Cin,Cout,stride', [
    (2, 16, 16, 128, 128, 1), (1, 32, 32, 256, 256, 1), (2, 9, 13, 128, 64, 1), (1, 16, 16, 128, 128, 2),
    (2, 7, 5, 256, 96, 2), (1, 4, 4, 768, 256, 2), (1, 1, 1, 128, 128, 2), (3, 2, 2, 128, 256, 1),
    (1, 32, 32, 128, 128, 2), (1, 64, 48, 256, 256, 1), (2, 32, 64, 256, 128, 1)])
@pytest.mark.parametrize('arith', ['bf16x3', 'f32'])
def test_conv3x3_nhwc_fwd_bwd_vs_torch(B, H, W, Cin, Cout, stride, arith):
    """both arithmetics of the implicit-GEMM convolutions (sm3_conv3x3_set_arith): the bf16x3 form (k-step 16; its weight
    gradient on the row-aligned gather when the output width is a multiple of 16) and the native fp32 MFMA form"""
    from sm3det_amd import _lib
    from sm3det_amd import _lib_backbone as LB
    from sm3det_amd.fpn import conv3x3_nhwc
    _lib.check(_lib.lib().sm3_conv3x3_set_arith({'bf16x3': 2, 'f32': 0}[arith]), 'conv3x3_set_arith')
    try:
        _conv_case(B, H, W, Cin, Cout, stride, conv3x3_nhwc)
    finally:
        _lib.lib().sm3_conv3x3_set_arith(LB.ARITH32)


def _conv_case(B, H, W, Cin, Cout, stride, conv3x3_nhwc):
    g = torch.Generator().manual_seed(B * 1000 + H * 10 + Cin)
    x = torch.randn(B, Cin, H, W, generator=g)
    w = torch.randn(Cout, Cin, 3, 3, generator=g) * 0.05
    b = torch.randn(Cout, generator=g)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    yr = F.conv2d(xr, wr, br, stride=stride, padding=1)
    go = torch.randn(yr.shape, generator=g)
    (yr * go).sum().backward()
    xd = x.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
    wd = w.permute(0, 2, 3, 1).contiguous().cuda().requires_grad_(True)
    bd = b.cuda().requires_grad_(True)
    y = conv3x3_nhwc(xd, wd, bd, stride)
    assert rel_err(y.permute(0, 3, 1, 2), yr) < FWD_TOL
    (y * go.permute(0, 2, 3, 1).cuda()).sum().backward()
    assert rel_err(xd.grad.permute(0, 3, 1, 2), xr.grad) < BWD_TOL
    assert rel_err(wd.grad.permute(0, 3, 1, 2), wr.grad) < BWD_TOL
    assert rel_err(bd.grad, br.grad) < BWD_TOL


def test_conv3x3_rejects_unsupported_channels():
    from sm3det_amd import _lib
    from sm3det_amd.fpn import conv3x3_nhwc
    x = torch.zeros(1, 4, 4, 48, device='cuda')
    with pytest.raises(_lib.SM3Error):
        conv3x3_nhwc(x, torch.zeros(64, 3, 3, 48, device='cuda'), torch.zeros(64, device='cuda'))


@pytest.mark.parametrize('B,H,W,C', [(2, 8, 8, 64), (1, 6, 10, 128), (3, 2, 2, 256)])
def test_upsample_add_and_gradient(B, H, W, C):
    from sm3det_amd.fpn import upsample2x_add
    g = torch.Generator().manual_seed(H + C)
    fine = torch.randn(B, H, W, C, generator=g).cuda().requires_grad_(True)
    coarse = torch.randn(B, H // 2, W // 2, C, generator=g).cuda().requires_grad_(True)
    out = upsample2x_add(fine, coarse)
    ref = fine.detach() + coarse.detach().repeat_interleave(2, 1).repeat_interleave(2, 2)
    assert torch.equal(out, ref)
    go = torch.randn(B, H, W, C, generator=g).cuda()
    (out * go).sum().backward()
    assert torch.equal(fine.grad, go)
    dc = go.view(B, H // 2, 2, W // 2, 2, C)
    dref = (dc[:, :, 0, :, 0] + dc[:, :, 0, :, 1]) + (dc[:, :, 1, :, 0] + dc[:, :, 1, :, 1])
    torch.testing.assert_close(coarse.grad, dref, rtol=1e-6, atol=1e-6)


def _ref_grads(net):
    out = {}
    for n, p in net.named_parameters():
        if p.grad is None:
            continue
        mod = net.get_submodule(n.rsplit('.', 1)[0])
        out[n] = mod._to_reference(p.grad) if n.endswith('weight') else p.grad
    return out


@pytest.mark.parametrize('name', CASES)
def test_module_matches_reference_fixture(name):
    from sm3det_amd.fpn import MultitaskFPN
    fx = load(name)
    kw = fx['cfg']
    net = MultitaskFPN(**kw)
    net.load_state_dict(MG.seeded_state_dict(fx['shapes']))
    net = net.cuda()
    for sl in fx['runs']:
        for p in net.parameters():
            p.grad = None
        # inputs as the backbone hands them over: logically NCHW over NHWC memory
        xs = [x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
              for x in MG.seeded_inputs(kw, fx['s0'], fx['batch'])]
        extra = kw.get('add_extra_convs', False)
        outs = net(xs, start_level=sl) if not extra else net(xs, start_level=sl, add_extra_convs=extra)
        sum((o * q.cuda()).sum() for o, q in zip(outs, MG.seeded_proj(outs, sl))).backward()
        check_run(fx, sl, outs, _ref_grads(net), [x.grad for x in xs], FWD_TOL, BWD_TOL)


def test_main_config_channels_vs_cpu_oracle():
    """main_SM3Det.py neck (in 96/192/384/768 -> 256, 5 outs, extra_level 1, 'on_output') on a 64x64-image pyramid,
    both call patterns of the detector (start_level 0 for the 2-stage branch, 1 for the 1-stage branch)."""
    from oracle import fpn_oracle as FO
    from sm3det_amd.fpn import MultitaskFPN
    kw = dict(in_channels=[96, 192, 384, 768], out_channels=256, extra_level=1, add_extra_convs='on_output', num_outs=5)
    net = MultitaskFPN(**kw)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    sd = MG.seeded_state_dict(shapes)
    net.load_state_dict(sd)
    net = net.cuda()
    for sl in (0, 1):
        for p in net.parameters():
            p.grad = None
        xs_cpu = MG.seeded_inputs(kw, 16, 2)
        p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
        xr = [x.clone().requires_grad_(True) for x in xs_cpu]
        ro = FO.fpn_forward(xr, p, num_outs=5, start_level=sl, add_extra_convs='on_output')
        proj = MG.seeded_proj(ro, sl)
        sum((o * q).sum() for o, q in zip(ro, proj)).backward()
        xs = [x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True) for x in xs_cpu]
        outs = net(xs, start_level=sl, add_extra_convs='on_output')
        sum((o * q.cuda()).sum() for o, q in zip(outs, proj)).backward()
        for o, r in zip(outs, ro):
            assert rel_err(o, r) < FWD_TOL
        grads = _ref_grads(net)
        for k, v in p.items():
            if v.grad is not None:
                assert rel_err(grads[k], v.grad) < BWD_TOL, k
        for a, b in zip(xs, xr):
            if b.grad is not None:
                assert rel_err(a.grad, b.grad) < BWD_TOL
