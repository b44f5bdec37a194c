"""CPU tier: the SAR branch's loss side (sm3det_amd/gfl_losses.py + GFLHead.loss: ATSS assignment, QFL / DFL / GIoU in a
fixed-shape masked form without host syncs) against oracle/gfl_oracle.py, a second restatement of the same mmdet 2.25
code in mmdet's own `nonzero()` / per-gt-loop control flow.  Both are restatements of code the reference does not vendor:
this pins the product on the oracle, not on mmdet itself (PARITY UNPINNED, stated in both modules)."""
import pytest
import torch

from oracle import gfl_oracle as GO
from sm3det_amd import gfl_losses as GL


def _gts(n, extent, seed, small=False):
    g = torch.Generator().manual_seed(seed)
    wh = torch.rand(n, 2, generator=g) * (extent / (8 if small else 3)) + 6
    c = torch.rand(n, 2, generator=g) * (extent - 40) + 20
    b = torch.cat([c - wh / 2, c + wh / 2], 1).clamp(0, extent)
    return b, torch.randint(0, 26, (n,), generator=g)


def _anchors(extent, strides):
    from sm3det_amd.rpn_head import grid_anchors
    sizes = [(extent // s, extent // s) for s in strides]
    return sizes, grid_anchors(sizes, strides, [8], [1.0], device='cpu')


@pytest.mark.parametrize('seed,ngt', [(0, 8), (1, 1), (2, 23), (3, 0)])
def test_atss_assign_matches_the_indexing_form(seed, ngt):
    strides = [8, 16, 32, 64, 128]
    _, lvl = _anchors(512, strides)
    flat = torch.cat(lvl)
    gtb, gtl = _gts(ngt, 512.0, seed, small=seed == 2)
    got = GL.atss_assign(flat, [a.shape[0] for a in lvl], gtb, gtl, topk=9)
    exp = GO.atss_assign(flat, [a.shape[0] for a in lvl], gtb, gtl, topk=9)
    assert torch.equal(got[0], exp[0])
    if ngt:
        assert torch.equal(got[1], exp[1])
        assert int((got[0] > 0).sum()) > 0
    assert torch.equal(got[2], exp[2])


def test_iou_forms_agree():
    a, _ = _gts(7, 100.0, 5)
    b, _ = _gts(5, 100.0, 6)
    assert torch.allclose(GL.bbox_overlaps(a, b), GO._iou_pairwise(a, b), atol=1e-7)
    assert torch.allclose(GL.bbox_overlaps(a, b), GO.iou_matrix(a, b), atol=0)
    g = torch.stack([GO._iou_one(x, y, 1e-7, giou=True) for x, y in zip(a[:5], b)])
    assert torch.allclose(GL.bbox_overlaps(a[:5], b, mode='giou', is_aligned=True, eps=1e-7), g, atol=1e-7)


@pytest.mark.parametrize('seed,ngts', [(0, (8, 8)), (1, (3, 0)), (2, (1, 17))])
def test_gfl_loss_matches_the_indexing_form_values_and_gradients(seed, ngts):
    from sm3det_amd.gfl_head import GFLHead
    torch.manual_seed(seed)
    strides = [8, 16, 32, 64, 128]
    extent = 256
    head = GFLHead(num_classes=26, in_channels=256, stacked_convs=1, feat_channels=32, reg_max=16,
                   norm_cfg=dict(type='GN', num_groups=32, requires_grad=True),
                   loss_cls=dict(type='QualityFocalLoss', use_sigmoid=True, beta=2.0, loss_weight=1.0),
                   loss_dfl=dict(type='DistributionFocalLoss', loss_weight=0.25), loss_bbox=dict(type='GIoULoss', loss_weight=2.0),
                   train_cfg=dict(assigner=dict(type='ATSSAssigner', topk=9), allowed_border=-1, pos_weight=-1, debug=False))
    sizes, lvl = _anchors(extent, strides)
    B = len(ngts)
    cls = [(torch.randn(B, 26, h, w) * 2 - 2).requires_grad_(True) for h, w in sizes]
    reg = [torch.randn(B, 68, h, w).requires_grad_(True) for h, w in sizes]
    gts = [_gts(n, float(extent), 10 * seed + i) for i, n in enumerate(ngts)]
    metas = [dict(img_shape=(extent, extent, 3), pad_shape=(extent, extent, 3)) for _ in range(B)]
    got = head.loss_torch(cls, reg, [g[0] for g in gts], [g[1] for g in gts], metas)  # the masked torch restatement (the kernels are tested against it on the GPU)
    exp = GO.gfl_loss([c.detach() for c in cls], [r.detach() for r in reg], lvl, strides, [g[0] for g in gts],
                      [g[1] for g in gts], 26)
    for key in ('loss_cls', 'loss_bbox', 'loss_dfl'):
        assert len(got[key]) == len(strides)
        for a, b in zip(got[key], exp[key]):
            assert abs(float(a) - float(b)) <= 1e-5 * max(1.0, abs(float(b))), (key, float(a), float(b))
    assert sum(float(v) for v in got['loss_bbox']) > 0
    # gradients: the oracle form through autograd on fresh leaves
    cls2 = [c.detach().clone().requires_grad_(True) for c in cls]
    reg2 = [r.detach().clone().requires_grad_(True) for r in reg]
    e2 = GO.gfl_loss(cls2, reg2, lvl, strides, [g[0] for g in gts], [g[1] for g in gts], 26)
    sum(sum(v) for v in e2.values()).backward()
    sum(sum(v) for v in got.values()).backward()
    for a, b in zip(cls + reg, cls2 + reg2):
        assert torch.allclose(a.grad, b.grad, atol=1e-6, rtol=1e-4)


def test_valid_flags_mask_out_anchors_beyond_the_padded_image():
    """pad_shape smaller than the feature maps' extent: the out-of-image anchors get weight 0 and are never positive"""
    from sm3det_amd.gfl_head import GFLHead
    head = GFLHead(num_classes=26, in_channels=256, stacked_convs=1, feat_channels=32, reg_max=16,
                   train_cfg=dict(assigner=dict(type='ATSSAssigner', topk=9), allowed_border=-1, pos_weight=-1))
    sizes, lvl = _anchors(256, [8, 16, 32, 64, 128])
    valid = torch.cat(head._valid_flags(sizes, (200, 256, 3), 'cpu'))
    assert int(valid.sum()) == 25 * 32 + 13 * 16 + 7 * 8 + 4 * 4 + 2 * 2
    gtb, gtl = _gts(6, 190.0, 3)
    lab, w, tgt, pos = head.get_targets(torch.cat(lvl), [a.shape[0] for a in lvl], [valid], [gtb], [gtl])
    assert not bool((pos[0] & ~valid).any()) and float(w[0][~valid].abs().sum()) == 0.0
    assert int(pos.sum()) > 0


def test_atss_with_partially_valid_levels_equals_assignment_on_the_compacted_anchors():
    """mmdet drops the anchors outside the padded image before ATSS runs (`num_level_anchors_inside`); the masked product
    form must give the same assignment even when a level keeps FEWER than topk valid anchors (its invalid anchors then sit
    among the level's top-k candidates and must not enter the mean + std threshold).  Checked against the indexing-form
    restatement (oracle/gfl_oracle.py) run on the compacted anchor list."""
    from oracle import gfl_oracle as GO
    from sm3det_amd.gfl_losses import atss_assign
    g = torch.Generator().manual_seed(11)
    levels = [64, 16, 4]
    anchors = []
    for li, nl in enumerate(levels):
        c = torch.rand(nl, 2, generator=g) * 120 + 4
        s = 8.0 * 2 ** li
        anchors.append(torch.cat([c - s / 2, c + s / 2], 1))
    bboxes = torch.cat(anchors)
    valid = torch.ones(sum(levels), dtype=torch.bool)
    valid[40:64] = False           # level 0: 40 valid
    valid[64 + 5:64 + 16] = False  # level 1: 5 valid (< topk = 9)
    valid[80 + 1:] = False         # level 2: 1 valid
    gts = torch.tensor([[10., 10., 60., 70.], [50., 40., 120., 110.], [5., 80., 40., 125.]])
    gtl = torch.tensor([3, 7, 1])
    gi, mo, lab = atss_assign(bboxes, levels, gts, gtl, topk=9, valid=valid)
    keep = valid.nonzero().squeeze(1)
    inside = [int(valid[:64].sum()), int(valid[64:80].sum()), int(valid[80:].sum())]
    gi_c, mo_c, lab_c = GO.atss_assign(bboxes[keep], inside, gts, gtl, topk=9)
    assert torch.equal(gi[keep], gi_c) and torch.equal(lab[keep], lab_c)
    assert int((gi[~valid] != 0).sum()) == 0 and int((gi > 0).sum()) > 0
    torch.testing.assert_close(mo[keep], mo_c)
