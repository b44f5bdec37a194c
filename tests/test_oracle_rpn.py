"""CPU tier: the RPN oracle (oracle/rpn_oracle.py) against the reference-generated fixture and, when /root/reference
exists, against the live reference functions (bit-exact: same torch ops in the same order); state_dict schema of the
MI355X head."""
import os

import pytest
import torch

from oracle import rpn_oracle as RO

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _fixture():
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import make_golden_rpn as MG
    return MG, torch.load(os.path.join(ROOT, 'tests', 'golden', 'rpn_decode.pt'), weights_only=False)


def test_decode_oracle_matches_reference_fixture():
    MG, fx = _fixture()
    a, d = MG.seeded_case(fx['n'])
    obb = RO.delta2bbox(a, d, fx['means'], fx['stds'])
    torch.testing.assert_close(obb, fx['proposals'], rtol=1e-6, atol=1e-5)
    torch.testing.assert_close(RO.obb2xyxy_le90(obb), fx['hboxes'], rtol=1e-6, atol=1e-4)
    assert torch.isfinite(obb[:, :4]).all()


def test_oracle_matches_live_reference_functions():
    from oracle import ref_rpn
    if not ref_rpn.available():
        pytest.skip('/root/reference not present (GPU box)')
    T, C, X = ref_rpn.load()
    MG, fx = _fixture()
    a, d = MG.seeded_case(3000)
    ref = C.delta2bbox(a, d, fx['means'], fx['stds'], 16 / 1000, 'le90')
    got = RO.delta2bbox(a, d, fx['means'], fx['stds'])
    assert torch.equal(ref, got)
    assert torch.equal(T.obb2xyxy(ref, 'le90'), RO.obb2xyxy_le90(got))
    assert torch.equal(T.poly2obb(torch.randn(64, 8, generator=torch.Generator().manual_seed(1)), 'le90'),
                       RO.poly2obb_le90(torch.randn(64, 8, generator=torch.Generator().manual_seed(1))))


def test_coder_oracles_match_fixture_and_live_reference():
    MG, fx = _fixture()
    props, gt, rois, deltas = MG.seeded_boxes()
    assert torch.equal(RO.midpoint_bbox2delta(props, gt, fx['means'], fx['stds']), fx['midpoint_encode'])
    for (es, pj), ref in fx['xywha'].items():
        assert torch.equal(RO.xywha_bbox2delta(rois, gt, MG.X_MEANS, MG.X_STDS, None, es, pj), ref['encode'])
        assert torch.equal(RO.xywha_delta2bbox(rois, deltas, MG.X_MEANS, MG.X_STDS, None, 16 / 1000, None, es, pj),
                           ref['decode'])
        assert torch.equal(RO.xywha_delta2bbox(rois, deltas, MG.X_MEANS, MG.X_STDS, (512, 640), 16 / 1000, None, es,
                                               pj), ref['decode_clamped'])
    from oracle import ref_rpn
    if ref_rpn.available():
        T, C, X = ref_rpn.load()
        assert torch.equal(C.bbox2delta(props, gt, fx['means'], fx['stds'], 'le90'), fx['midpoint_encode'])
        assert torch.equal(T.obb2poly(gt, 'le90'), RO.obb2poly_le90(gt))


def test_proposals_oracle_properties():
    """get_bboxes_single: scores descending, at most max_per_img, per-level NMS independence (levels far apart in
    the offset space never suppress each other)."""
    g = torch.Generator().manual_seed(5)
    A, sizes, strides = 3, [(8, 8), (4, 4)], [8, 16]
    from sm3det_amd.rpn_head import grid_anchors
    anchors = grid_anchors(sizes, strides, device='cpu')
    cls = [torch.randn(A, h, w, generator=g) for h, w in sizes]
    reg = [torch.randn(6 * A, h, w, generator=g) * 0.3 for h, w in sizes]
    cfg = dict(nms_pre=100, max_per_img=50, nms=dict(type='nms', iou_threshold=0.8), min_bbox_size=0)
    dets = RO.get_bboxes_single(cls, reg, anchors, cfg, (0.,) * 6, (1., 1., 1., 1., 0.5, 0.5))
    assert dets.shape[1] == 6 and 0 < dets.shape[0] <= 50
    assert (dets[1:, 5] <= dets[:-1, 5]).all()
    assert anchors[0].shape == (8 * 8 * A, 4) and anchors[1].shape == (4 * 4 * A, 4)
    # centre of the first anchor cell is (0, 0); ratio 0.5 is the wide one (w = 2h)
    w0, h0 = anchors[0][0, 2] - anchors[0][0, 0], anchors[0][0, 3] - anchors[0][0, 1]
    assert abs(float(w0 / h0) - 2.0) < 1e-5 and abs(float(anchors[0][0, :2].sum() + anchors[0][0, 2:].sum())) < 1e-4


def test_head_state_dict_schema():
    from sm3det_amd.rpn_head import OrientedRPNHead
    head = OrientedRPNHead(in_channels=256, feat_channels=256, version='le90',
                           anchor_generator=dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0],
                                                 strides=[4, 8, 16, 32, 64]),
                           bbox_coder=dict(type='MidpointOffsetCoder', angle_range='le90',
                                           target_means=[0.0] * 6, target_stds=[1.0, 1.0, 1.0, 1.0, 0.5, 0.5]),
                           loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0))
    sd = head.state_dict()
    assert {k: tuple(v.shape) for k, v in sd.items()} == {
        'rpn_conv.weight': (256, 256, 3, 3), 'rpn_conv.bias': (256,), 'rpn_cls.weight': (3, 256, 1, 1),
        'rpn_cls.bias': (3,), 'rpn_reg.weight': (18, 256, 1, 1), 'rpn_reg.bias': (18,)}
    ref = {k: torch.randn(v.shape, generator=torch.Generator().manual_seed(i)) for i, (k, v) in enumerate(sd.items())}
    head.load_state_dict(ref)
    for k, v in head.state_dict().items():
        assert torch.equal(v, ref[k])
    head.init_weights()
    assert float(head.rpn_cls.bias.abs().max()) == 0.0


def test_host_side_roi_helpers_on_cpu():
    """map_roi_levels (rotate_single_level_roi_extractor.py:66-84) and rbbox2roi are plain torch host logic."""
    from oracle import roi_oracle
    from sm3det_amd.roi_head import RotatedShared2FCBBoxHead, RotatedSingleRoIExtractor
    from sm3det_amd.rpn_head import rbbox2roi
    ext = RotatedSingleRoIExtractor(dict(type='RoIAlignRotated', out_size=7, sample_num=2, clockwise=True), 256,
                                    [4, 8, 16, 32])
    assert ext.num_inputs == 4 and ext.output_size == (7, 7) and ext.sampling_ratio == 2 and ext.clockwise
    g = torch.Generator().manual_seed(2)
    rois = torch.rand(500, 6, generator=g)
    rois[:, 3:5] = torch.exp(rois[:, 3:5] * 5 + 1.5)
    lv = ext.map_roi_levels(rois, 4)
    assert torch.equal(lv, roi_oracle.map_roi_levels(rois, 4)) and set(lv.tolist()) == {0, 1, 2, 3}
    # level boundaries of the FPN paper rule with finest_scale 56
    edge = torch.tensor([[0, 0, 0, 111.9, 111.9, 0], [0, 0, 0, 112.1, 112.1, 0], [0, 0, 0, 447.9, 447.9, 0],
                         [0, 0, 0, 448.1, 448.1, 0], [0, 0, 0, 5000.0, 5000.0, 0]])
    assert ext.map_roi_levels(edge, 4).tolist() == [0, 1, 2, 3, 3]
    r = rbbox2roi([torch.ones(2, 6), torch.zeros(0, 6), torch.full((1, 5), 2.0)])
    assert tuple(r.shape) == (3, 6) and r[:, 0].tolist() == [0.0, 0.0, 2.0] and r[2, 1:].tolist() == [2.0] * 5
    with pytest.raises(NotImplementedError):
        RotatedSingleRoIExtractor(dict(type='RoIAlign'), 256, [4])
    with pytest.raises(NotImplementedError):
        RotatedShared2FCBBoxHead(with_avg_pool=True)
    head = RotatedShared2FCBBoxHead(num_classes=26, reg_class_agnostic=False)
    assert head.fc_reg.out_features == 130 and head.fc_cls.out_features == 27


def test_coder_round_trips_on_the_oracle():
    """size-independent property: decode(encode(gt)) == gt for both coders (regular-form boxes)"""
    from tests.fpn_common import assert_boxes_close, coder_round_trip_cases
    gt, anchors, rois = coder_round_trip_cases()
    stds6 = (1., 1., 1., 1., 0.5, 0.5)
    back = RO.delta2bbox(anchors, RO.midpoint_bbox2delta(anchors, gt, (0.,) * 6, stds6), (0.,) * 6, stds6)
    assert_boxes_close(back, gt, 1e-3, 1e-5)
    m5, s5 = (0.,) * 5, (0.1, 0.1, 0.2, 0.2, 0.1)
    for es, pj in ((True, True), (False, False)):
        d = RO.xywha_bbox2delta(rois, gt, m5, s5, None, es, pj)
        assert_boxes_close(RO.xywha_delta2bbox(rois, d, m5, s5, None, 16 / 1000, None, es, pj), gt, 5e-4, 1e-5)
