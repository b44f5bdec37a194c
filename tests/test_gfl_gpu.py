"""GPU tier: GFL head conv towers (sm3det_amd/gfl_head.py: implicit-GEMM 3x3 convs + GroupNorm/ReLU kernels) against
the plain-torch restatement oracle/gfl_oracle.py (mmdet semantics -- PARITY UNPINNED, see that file) and GroupNorm
against torch.nn.functional.group_norm in fp64.  Tolerances: forward 1e-4, gradients 1e-3 (max-norm relative)."""
import pytest
import torch
import torch.nn.functional as F

from tests.moe_common import rel_err

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('B,H,W,C,G,relu', [(2, 16, 24, 256, 32, True), (1, 8, 8, 256, 32, False), (3, 5, 7, 96, 8, True),
                                            (1, 64, 64, 256, 32, True), (2, 3, 3, 512, 32, True)])
def test_groupnorm_relu_fwd_bwd_vs_torch_fp64(B, H, W, C, G, relu):
    from sm3det_amd.gfl_head import group_norm_relu
    g = torch.Generator().manual_seed(C + H)
    x = (torch.randn(B, H, W, C, generator=g) * 2 + 0.5).cuda().requires_grad_(True)
    w = (torch.rand(C, generator=g) + 0.5).cuda().requires_grad_(True)
    b = torch.randn(C, generator=g).cuda().requires_grad_(True)
    y = group_norm_relu(x, w, b, G, 1e-5, relu)
    go = torch.randn(B, H, W, C, generator=g).cuda()
    y.backward(go)
    xr, wr, br = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    yr = F.group_norm(xr.permute(0, 3, 1, 2), G, wr, br, 1e-5)
    if relu:
        yr = F.relu(yr)
    yr = yr.permute(0, 2, 3, 1)
    yr.backward(go.double())
    assert rel_err(y, yr) < 1e-5
    assert rel_err(x.grad, xr.grad) < 1e-4 and rel_err(w.grad, wr.grad) < 1e-4 and rel_err(b.grad, br.grad) < 1e-4


def test_gfl_head_config_shapes_state_dict_and_values_vs_oracle():
    from oracle import gfl_oracle as GO
    from sm3det_amd.gfl_head import GFLHead
    torch.manual_seed(0)
    head = GFLHead(num_classes=26, in_channels=256, stacked_convs=4, feat_channels=256,
                   anchor_generator=dict(type='AnchorGenerator', ratios=[1.0], octave_base_scale=8,
                                         scales_per_octave=1, strides=[8, 16, 32, 64, 128]),
                   loss_cls=dict(type='QualityFocalLoss', use_sigmoid=True, beta=2.0, loss_weight=1.0),
                   loss_dfl=dict(type='DistributionFocalLoss', loss_weight=0.25), reg_max=16,
                   loss_bbox=dict(type='GIoULoss', loss_weight=2.0))  # local_configs/main_SM3Det.py:29-48
    sd = head.state_dict()
    assert sd['cls_convs.0.conv.weight'].shape == (256, 256, 3, 3) and 'cls_convs.0.conv.bias' not in sd
    assert sd['cls_convs.3.gn.weight'].shape == (256,) and sd['gfl_cls.weight'].shape == (26, 256, 3, 3)
    assert sd['gfl_reg.weight'].shape == (68, 256, 3, 3) and sd['scales.4.scale'].shape == ()
    g = torch.Generator().manual_seed(1)
    with torch.no_grad():  # O(1) weights so the comparison is not dominated by the 0.01-std init
        for n, p in head.named_parameters():
            if n.endswith('conv.weight') or n.endswith('gfl_cls.weight') or n.endswith('gfl_reg.weight'):
                p.copy_(torch.randn(p.shape, generator=g) * (2.0 / (9 * 256)) ** 0.5)
            elif n.endswith('gn.weight'):
                p.copy_(torch.rand(p.shape, generator=g) + 0.5)
            elif n.endswith('bias'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.1)
            elif n.endswith('scale'):
                p.copy_(torch.rand((), generator=g) + 0.5)
    ref_p = {k: v.detach().double().requires_grad_(True) for k, v in head.state_dict().items()}
    head = head.cuda()
    feats = [torch.randn(2, 256, s, s, generator=g) for s in (16, 8, 4, 2, 1)]  # 5 levels of a 128^2 image
    fg = [f.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True) for f in feats]
    cls, reg = head(fg)
    fr = [f.double().requires_grad_(True) for f in feats]
    rc, rr = GO.forward(fr, ref_p)
    for a, b in zip(cls + reg, rc + rr):
        assert tuple(a.shape) == tuple(b.shape)
        assert rel_err(a, b) < 1e-4, rel_err(a, b)
    R = [torch.randn(t.shape, generator=g) for t in rc + rr]
    sum((a * r.cuda()).sum() for a, r in zip(cls + reg, R)).backward()
    sum((a * r.double()).sum() for a, r in zip(rc + rr, R)).backward()
    got = dict(head.state_dict(keep_vars=True))
    worst = (0.0, None)
    for k, v in ref_p.items():
        gp = dict(head.named_parameters())[k].grad
        if k.endswith('conv.weight') or k in ('gfl_cls.weight', 'gfl_reg.weight'):
            gp = gp.permute(0, 3, 1, 2)  # kernel layout (Cout,3,3,Cin) -> reference (Cout,Cin,3,3)
        e = rel_err(gp, v.grad)
        if e > worst[0]:
            worst = (e, k)
    assert worst[0] < 1e-3, worst
    for a, b in zip(fg, fr):
        assert rel_err(a.grad, b.grad) < 1e-3
    assert len(got) == len(ref_p)
