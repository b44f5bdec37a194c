"""GPU tier: fused multi-level RoI extractor and the Shared2FC box head against the CPU oracle."""
import pytest
import torch

from tests.fpn_common import rel_err

pytestmark = pytest.mark.gpu


def _rois(n, batch, extent, g):
    r = torch.zeros(n, 6)
    r[:, 0] = torch.randint(0, batch, (n,), generator=g).float()
    r[:, 1:3] = torch.rand(n, 2, generator=g) * extent
    # sizes spread over all four FPN levels (finest_scale 56: < 112, < 224, < 448, >= 448)
    r[:, 3] = torch.exp(torch.rand(n, generator=g) * 4.2 + 2.0)
    r[:, 4] = r[:, 3] * (0.3 + torch.rand(n, generator=g))
    r[:, 5] = (torch.rand(n, generator=g) - 0.5) * 3.1
    return r


@pytest.mark.parametrize('channels_last', [True, False])
def test_multilevel_extractor_matches_oracle(channels_last):
    from oracle import roi_oracle as RO
    from sm3det_amd.roi_head import RotatedSingleRoIExtractor
    g = torch.Generator().manual_seed(31)
    B, C, strides = 2, 32, [4, 8, 16, 32]
    feats = [torch.randn(B, C, 256 // s, 256 // s, generator=g) for s in strides]
    rois = _rois(300, B, 256.0, g)
    ext = RotatedSingleRoIExtractor(dict(type='RoIAlignRotated', out_size=7, sample_num=2, clockwise=True), C, strides)
    ref, lv = RO.extract(feats, rois, strides)
    assert len(torch.unique(lv)) == 4  # every level is exercised
    fmt = torch.channels_last if channels_last else torch.contiguous_format
    fd = [f.cuda().contiguous(memory_format=fmt).requires_grad_(True) for f in feats]
    out, levels = ext(fd, rois.cuda(), return_levels=True)
    assert torch.equal(levels.cpu().long(), lv)
    assert torch.equal(ext.map_roi_levels(rois, 4), lv)
    assert rel_err(out, ref) < 1e-5
    gout = torch.randn(out.shape, generator=g)
    out.backward(gout.cuda())
    gref = RO.extract_backward(gout, [tuple(f.shape) for f in feats], rois, strides)
    for a, b in zip(fd, gref):
        assert rel_err(a.grad, b) < 1e-4
    # empty RoI list
    e = ext([f.detach() for f in fd], torch.zeros(0, 6, device='cuda'))
    assert tuple(e.shape) == (0, C, 7, 7)


def test_shared2fc_head_forward_backward_vs_oracle():
    from oracle import roi_oracle as RO
    from sm3det_amd.roi_head import RotatedShared2FCBBoxHead
    g = torch.Generator().manual_seed(8)
    head = RotatedShared2FCBBoxHead(in_channels=256, fc_out_channels=1024, roi_feat_size=7, num_classes=26,
                                    reg_class_agnostic=True)
    sd = {k: torch.randn(v.shape, generator=g) * (0.02 if v.dim() > 1 else 0.1) for k, v in head.state_dict().items()}
    head.load_state_dict(sd)
    head = head.cuda()
    x = torch.randn(96, 256, 7, 7, generator=g)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    rc, rr = RO.shared2fc_forward(xr, p)
    gc, gr = torch.randn(rc.shape, generator=g), torch.randn(rr.shape, generator=g)
    ((rc * gc).sum() + (rr * gr).sum()).backward()
    xd = x.cuda().requires_grad_(True)
    cls, reg = head(xd)
    assert tuple(cls.shape) == (96, 27) and tuple(reg.shape) == (96, 5)
    assert rel_err(cls, rc) < 1e-4 and rel_err(reg, rr) < 1e-4
    ((cls * gc.cuda()).sum() + (reg * gr.cuda()).sum()).backward()
    assert rel_err(xd.grad, xr.grad) < 1e-3
    for n, q in head.named_parameters():
        assert rel_err(q.grad, p[n].grad) < 1e-3, n
