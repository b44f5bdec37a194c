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


# ------------------------------------------------------------------------------------------------ test-time path (g2)
@pytest.mark.parametrize('n,nc,seed', [(400, 5, 0), (1500, 26, 1), (60, 3, 2)])
def test_multiclass_nms_rotated_matches_oracle(n, nc, seed):
    """sm3det_amd.post_processing.multiclass_nms_rotated on the GPU (class-offset trick + the sm3_nms_rotated kernel) vs
    oracle/roi_oracle.py (pinned on the reference's own function run live, tests/test_oracle_heads_live.py): detections
    bit-equal, labels and kept candidate indices equal; score filter, max_num cut and the empty case."""
    import numpy as np
    from oracle import roi_oracle as RO
    from sm3det_amd.post_processing import multiclass_nms_rotated
    from tests import synth
    g = torch.Generator().manual_seed(seed)
    boxes = torch.from_numpy(synth.rotated_boxes(n, seed, cluster=True))
    scores = torch.softmax(torch.randn(n, nc + 1, generator=g) * 2.0, -1)
    for thr, max_num in ((0.05, 2000), (0.2, 37), (0.999, 100)):
        d, l, k = multiclass_nms_rotated(boxes.cuda(), scores.cuda(), thr, dict(iou_thr=0.1), max_num, return_inds=True)
        dr, lr, kr = RO.multiclass_nms_rotated(boxes.numpy(), scores.numpy(), thr, 0.1, max_num)
        assert np.array_equal(d.cpu().numpy(), dr) and np.array_equal(l.cpu().numpy(), lr)
        if dr.shape[0]:
            assert np.array_equal(k.cpu().numpy(), kr)
    assert dr.shape[0] == 0 and tuple(d.shape) == (0, 6)


@pytest.mark.parametrize('rescale', [False, True])
def test_roi_head_simple_test_matches_oracle(rescale):
    """OrientedStandardRoIHead.simple_test of the product (RoI extractor kernel -> Shared2FC GEMMs -> softmax ->
    sm3_delta_xywha_decode_le90 -> rescale -> multiclass rotated NMS -> rbbox2result) vs the CPU oracle composition that
    the reference's own classes pin: per image and class the same number of detections, boxes <= 1e-4 px / 1e-5 rad,
    scores <= 1e-6."""
    import numpy as np
    from oracle import roi_oracle as RO
    from sm3det_amd.roi_head import OrientedStandardRoIHead
    C, strides = 6, [4, 8, 16, 32]
    means, stds = (0.,) * 5, (0.1, 0.1, 0.2, 0.2, 0.1)
    cfg = dict(nms_pre=2000, min_bbox_size=0, score_thr=0.05, nms=dict(iou_thr=0.1), max_per_img=2000)
    head = OrientedStandardRoIHead(
        bbox_roi_extractor=dict(type='RotatedSingleRoIExtractor',
                                roi_layer=dict(type='RoIAlignRotated', out_size=7, sample_num=2, clockwise=True),
                                out_channels=32, featmap_strides=strides),
        bbox_head=dict(type='RotatedShared2FCBBoxHead', in_channels=32, fc_out_channels=64, roi_feat_size=7,
                       num_classes=C, reg_class_agnostic=True,
                       bbox_coder=dict(type='DeltaXYWHAOBBoxCoder', angle_range='le90', norm_factor=None, edge_swap=True,
                                       proj_xy=True, target_means=means, target_stds=stds)),
        test_cfg=cfg, version='le90')
    g = torch.Generator().manual_seed(5)
    sd = {k: torch.randn(v.shape, generator=g) * (0.02 if v.dim() > 1 else 0.1) for k, v in head.bbox_head.state_dict().items()}
    sd['fc_cls.weight'] = sd['fc_cls.weight'] * 6.0
    sd['fc_reg.weight'] = sd['fc_reg.weight'] * 3.0
    head.bbox_head.load_state_dict(sd)
    head = head.cuda().eval()
    feats = [torch.randn(2, 32, 64 >> i, 64 >> i, generator=g) for i in range(4)]
    props = []
    for i in range(2):
        ctr = torch.rand(30, 2, generator=g) * 200 + 28
        c = ctr[torch.randint(0, 30, (150,), generator=g)] + torch.randn(150, 2, generator=g) * 6
        w = torch.exp(torch.rand(150, generator=g) * 3.0 + 2.2)
        props.append(torch.stack([c[:, 0], c[:, 1], w, w * (0.3 + 0.6 * torch.rand(150, generator=g)),
                                  (torch.rand(150, generator=g) - 0.5) * 3.0, torch.rand(150, generator=g)], 1))
    metas = [dict(img_shape=(256, 256, 3), scale_factor=np.array([1.25, 1.25, 1.25, 1.25], np.float32)) for _ in range(2)]
    exp = RO.simple_test(feats, props, metas, sd, strides, C, cfg, means, stds, dict(edge_swap=True, proj_xy=True),
                         rescale=rescale)
    got = head.simple_test([f.cuda() for f in feats], [p.cuda() for p in props], metas, rescale=rescale)
    assert len(got) == 2
    total = 0
    for gi, ei in zip(got, exp):
        assert len(gi) == len(ei) == C
        for a, b in zip(gi, ei):
            assert a.dtype == np.float32 and a.shape == b.shape, (a.shape, b.shape)
            if a.shape[0]:
                np.testing.assert_allclose(a[:, :4], b[:, :4], atol=1e-4, rtol=1e-5)
                np.testing.assert_allclose(a[:, 4], b[:, 4], atol=1e-5)
                np.testing.assert_allclose(a[:, 5], b[:, 5], atol=1e-6)
            total += a.shape[0]
    assert total > 20 and sum(1 for a in got[0] if a.shape[0]) >= 2
    # raw form (cfg None): decoded boxes and softmax scores of every proposal
    db, ds = head.simple_test_bboxes([f.cuda() for f in feats], metas, [p.cuda() for p in props], None, rescale=rescale)
    assert tuple(db[0].shape) == (150, 5) and tuple(ds[0].shape) == (150, C + 1)
