"""GPU tier: Oriented-RPN conv tower and proposal glue against the CPU oracle and the reference-generated fixture.
Tolerances: decode coordinates 1e-5 relative (+1e-3 px absolute), angles 1e-4 rad (mod pi); tower forward 1e-4,
gradients 1e-3 (max-norm relative); proposal lists: identical length and order, scores 1e-6."""
import math
import os

import pytest
import torch

from tests.fpn_common import rel_err

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STDS = (1., 1., 1., 1., 0.5, 0.5)


def _fixture():
    import sys
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    import make_golden_rpn as MG
    return MG, torch.load(os.path.join(ROOT, 'tests', 'golden', 'rpn_decode.pt'), weights_only=False)


def _check_obb(got, ref):
    torch.testing.assert_close(got[:, :4], ref[:, :4], rtol=1e-5, atol=2e-3)
    square = (ref[:, 2] - ref[:, 3]).abs() <= 1e-3 * ref[:, 2].abs().clamp_min(1e-6)  # edge1 ~ edge2: the angle branch
    d = (got[:, 4] - ref[:, 4] + math.pi / 2) % math.pi - math.pi / 2                  # may flip by rounding
    assert float(d[~square].abs().max()) < 1e-4
    assert int(square.sum()) < ref.shape[0] // 4


def test_decode_matches_reference_fixture_and_oracle():
    from oracle import rpn_oracle as RO
    from sm3det_amd.rpn_head import MidpointOffsetCoder
    MG, fx = _fixture()
    a, d = MG.seeded_case(fx['n'])
    coder = MidpointOffsetCoder(target_means=fx['means'], target_stds=fx['stds'], angle_range='le90')
    props, hb = coder.decode(a.cuda(), d.cuda())
    _check_obb(props.cpu(), fx['proposals'])
    ok = (fx['proposals'][:, 2] - fx['proposals'][:, 3]).abs() > 1e-3 * fx['proposals'][:, 2].abs()
    torch.testing.assert_close(hb.cpu()[ok], fx['hboxes'][ok], rtol=1e-5, atol=5e-3)
    # permutation prefix + score gather (the top-k path)
    g = torch.Generator().manual_seed(3)
    order = torch.randperm(fx['n'], generator=g)[:777]
    sc = torch.rand(fx['n'], generator=g)
    p2, h2, s2 = coder.decode(a.cuda(), d.cuda(), order=order.cuda(), scores=sc.cuda())
    assert torch.equal(p2, props[order.cuda()]) and torch.equal(h2, hb[order.cuda()])
    assert torch.equal(s2.cpu(), sc[order])
    _check_obb(p2.cpu(), RO.delta2bbox(a[order], d[order], fx['means'], fx['stds']))


def test_sigmoid_and_relu_bwd():
    from sm3det_amd import _lib_backbone as LB
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(10007, generator=g) * 6).cuda()
    y = torch.empty_like(x)
    LB.call('sigmoid_f32', x, y, x.numel())
    torch.testing.assert_close(y.cpu(), torch.sigmoid(x.cpu()), rtol=1e-6, atol=1e-7)
    dy, v = torch.randn(4096, generator=g).cuda(), torch.randn(4096, generator=g).cuda().clamp_min(0)
    dx = torch.empty_like(dy)
    LB.call('relu_bwd', dy, v, dx, dy.numel())
    assert torch.equal(dx, torch.where(v > 0, dy, torch.zeros_like(dy)))


def _head(feat=256):
    from sm3det_amd.rpn_head import OrientedRPNHead
    return OrientedRPNHead(in_channels=feat, feat_channels=feat, version='le90',
                           anchor_generator=dict(type='AnchorGenerator', scales=[8], ratios=[0.5, 1.0, 2.0],
                                                 strides=[4, 8, 16, 32, 64]),
                           bbox_coder=dict(type='MidpointOffsetCoder', angle_range='le90', target_means=[0.0] * 6,
                                           target_stds=list(STDS)),
                           loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
                           test_cfg=dict(nms_pre=2000, max_per_img=2000, nms=dict(type='nms', iou_threshold=0.8),
                                         min_bbox_size=0))


@pytest.mark.parametrize('B,H,W', [(2, 16, 16), (1, 9, 13), (2, 4, 4)])
def test_tower_forward_backward_vs_oracle(B, H, W):
    from oracle import rpn_oracle as RO
    head = _head()
    g = torch.Generator().manual_seed(H)
    sd = {k: torch.randn(v.shape, generator=g) * (0.03 if v.dim() > 1 else 0.1) for k, v in head.state_dict().items()}
    head.load_state_dict(sd)
    head = head.cuda()
    x = torch.randn(B, 256, H, W, generator=g)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    xr = x.clone().requires_grad_(True)
    rc, rr = RO.rpn_forward_single(xr, p)
    gc, gr = torch.randn(rc.shape, generator=g), torch.randn(rr.shape, generator=g)
    ((rc * gc).sum() + (rr * gr).sum()).backward()
    xd = x.cuda().contiguous(memory_format=torch.channels_last).requires_grad_(True)
    cls, reg = head.forward_single(xd)
    assert tuple(cls.shape) == tuple(rc.shape) and tuple(reg.shape) == tuple(rr.shape)
    assert rel_err(cls, rc) < 1e-4 and rel_err(reg, rr) < 1e-4
    ((cls * gc.cuda()).sum() + (reg * gr.cuda()).sum()).backward()
    assert rel_err(xd.grad, xr.grad) < 1e-3
    for n, q in head.named_parameters():
        mod = head.get_submodule(n.rsplit('.', 1)[0])
        got = mod._to_reference(q.grad) if n.endswith('weight') else q.grad
        assert rel_err(got, p[n].grad) < 1e-3, n


def test_proposals_match_oracle():
    """OrientedRPNHead._get_bboxes_single on a 5-level pyramid with nms_pre smaller than the largest levels."""
    from oracle import rpn_oracle as RO
    from sm3det_amd.rpn_head import grid_anchors
    head = _head().cuda()
    g = torch.Generator().manual_seed(21)
    sizes, strides, A = [(32, 32), (16, 16), (8, 8), (4, 4), (2, 2)], [4, 8, 16, 32, 64], 3
    cls = [torch.randn(A, h, w, generator=g) * 2 for h, w in sizes]
    reg = [torch.randn(6 * A, h, w, generator=g) * 0.4 for h, w in sizes]
    cfg = dict(nms_pre=600, max_per_img=500, nms=dict(type='nms', iou_threshold=0.8), min_bbox_size=0)
    anchors = grid_anchors(sizes, strides, device='cpu')
    ref = RO.get_bboxes_single(cls, reg, anchors, cfg, (0.,) * 6, STDS)
    got = head._get_bboxes_single([c.cuda() for c in cls], [r.cuda() for r in reg], [a.cuda() for a in anchors],
                                  (128, 128, 3), None, cfg).cpu()
    assert got.shape == ref.shape, (got.shape, ref.shape)
    torch.testing.assert_close(got[:, 5], ref[:, 5], rtol=1e-6, atol=1e-7)
    _check_obb(got[:, :5], ref[:, :5])
    assert (got[1:, 5] <= got[:-1, 5]).all()
    # batch front-end: anchors generated on the device, one list entry per image
    outs = head.get_bboxes([c[None].cuda() for c in cls], [r[None].cuda() for r in reg], cfg=cfg)
    assert len(outs) == 1 and torch.equal(outs[0].cpu(), got)


def test_gather_dict_values_pinned_upload():
    """sm3det_amd.h2d.gather_dict_values == the reference's torch.stack(...).cuda() / [t.cuda() ...] results."""
    from sm3det_amd.h2d import PinnedUploader, gather_dict_values
    g = torch.Generator().manual_seed(9)
    data = [dict(sar=torch.randn(3, 64, 64, generator=g), rgb=torch.randn(3, 64, 64, generator=g)) for _ in range(2)]
    data.append(dict(sar=torch.randn(3, 64, 64, generator=g)))
    up = PinnedUploader()
    for _ in range(2):  # second round re-uses the pinned staging buffers
        out = gather_dict_values(data, ['sar', 'rgb'], uploader=up)
        assert out['sar'].is_cuda and tuple(out['sar'].shape) == (3, 3, 64, 64)
        assert torch.equal(out['sar'].cpu(), torch.stack([d['sar'] for d in data]))
        assert torch.equal(out['rgb'].cpu(), torch.stack([d['rgb'] for d in data[:2]]))
    lst = gather_dict_values(data, ['sar'], ignore_tensor=True, uploader=up)['sar']
    assert isinstance(lst, list) and len(lst) == 3 and torch.equal(lst[2].cpu(), data[2]['sar'])
    assert all(s[0].is_pinned() for s in up._staging.values())


def test_box_coders_match_reference_fixture():
    """MidpointOffsetCoder.encode and DeltaXYWHAOBBoxCoder.encode/decode kernels vs the reference-generated fixture."""
    from sm3det_amd.rpn_head import DeltaXYWHAOBBoxCoder, MidpointOffsetCoder, rbbox2roi
    MG, fx = _fixture()
    props, gt, rois, deltas = MG.seeded_boxes()
    mc = MidpointOffsetCoder(target_means=fx['means'], target_stds=fx['stds'], angle_range='le90')
    torch.testing.assert_close(mc.encode(props.cuda(), gt.cuda()).cpu(), fx['midpoint_encode'], rtol=2e-5, atol=2e-5)
    for (es, pj), ref in fx['xywha'].items():
        xc = DeltaXYWHAOBBoxCoder(target_means=MG.X_MEANS, target_stds=MG.X_STDS, angle_range='le90', edge_swap=es,
                                  proj_xy=pj)
        enc = xc.encode(rois.cuda(), gt.cuda()).cpu()
        # the angle target wraps at +-pi/2: compare modulo the wrap (in units of the std-normalised delta)
        torch.testing.assert_close(enc[:, :4], ref['encode'][:, :4], rtol=2e-5, atol=2e-4)
        wrap = math.pi / MG.X_STDS[4]
        d = (enc[:, 4] - ref['encode'][:, 4] + wrap / 2) % wrap - wrap / 2
        assert float(d.abs().max()) < 1e-3
        for key, ms in (('decode', None), ('decode_clamped', (512, 640))):
            got = xc.decode(rois.cuda(), deltas.cuda(), max_shape=ms).cpu()
            near_square = (ref[key][:, 2] - ref[key][:, 3]).abs() < 1e-3 * ref[key][:, 2]
            torch.testing.assert_close(got[~near_square, :4], ref[key][~near_square, :4], rtol=2e-5, atol=2e-3)
            dd = (got[:, 4] - ref[key][:, 4] + math.pi / 2) % math.pi - math.pi / 2
            assert float(dd[~near_square].abs().max()) < 1e-4
    r = rbbox2roi([gt[:3].cuda(), gt[:0].cuda(), gt[3:5].cuda()])
    assert tuple(r.shape) == (5, 6) and r[:, 0].tolist() == [0.0, 0.0, 0.0, 2.0, 2.0]


def test_coder_round_trips_on_the_device():
    """decode(encode(gt)) == gt through the kernels (10x the oracle's own round-trip error as tolerance)"""
    from sm3det_amd.rpn_head import DeltaXYWHAOBBoxCoder, MidpointOffsetCoder
    from tests.fpn_common import assert_boxes_close, coder_round_trip_cases
    gt, anchors, rois = coder_round_trip_cases()
    mc = MidpointOffsetCoder(target_means=[0.0] * 6, target_stds=list(STDS), angle_range='le90')
    back, _ = mc.decode(anchors.cuda(), mc.encode(anchors.cuda(), gt.cuda()))
    assert_boxes_close(back.cpu(), gt, 1e-2, 1e-4)
    for es, pj in ((True, True), (False, False)):
        xc = DeltaXYWHAOBBoxCoder(target_means=[0.0] * 5, target_stds=[0.1, 0.1, 0.2, 0.2, 0.1], angle_range='le90',
                                  edge_swap=es, proj_xy=pj)
        back = xc.decode(rois.cuda(), xc.encode(rois.cuda(), gt.cuda()))
        assert_boxes_close(back.cpu(), gt, 5e-3, 1e-4)


def test_get_bboxes_fixed_equals_the_variable_length_path_and_is_graph_capturable():
    """sync-free fixed-size proposals == the first `count` rows of `_get_bboxes_single`, zeros after; and the call can
    be captured in a hipGraph (no host synchronisation inside)."""
    from sm3det_amd.rpn_head import OrientedRPNHead, grid_anchors
    torch.manual_seed(0)
    rpn = OrientedRPNHead(in_channels=256, feat_channels=256, version='le90',
                          bbox_coder=dict(type='MidpointOffsetCoder', angle_range='le90', target_means=[0.0] * 6,
                                          target_stds=[1.0, 1.0, 1.0, 1.0, 0.5, 0.5])).cuda()
    rpn.init_weights()
    g = torch.Generator().manual_seed(3)
    sizes = [(64, 64), (32, 32), (16, 16), (8, 8), (4, 4)]
    cls = [torch.randn(2, 3, h, w, generator=g).cuda() for h, w in sizes]
    reg = [torch.randn(2, 18, h, w, generator=g).cuda() * 0.5 for h, w in sizes]
    anchors = grid_anchors(sizes, [4, 8, 16, 32, 64], device='cuda')
    cfg = dict(nms_pre=1000, max_per_img=300, nms=dict(type='nms', iou_threshold=0.8), min_bbox_size=0)
    props, counts = rpn.get_bboxes_fixed(cls, reg, (256, 256, 3), cfg, mlvl_anchors=anchors)
    assert props.shape == (2, 300, 6) and counts.dtype == torch.int32
    for i in range(2):
        ref = rpn._get_bboxes_single([c[i] for c in cls], [r[i] for r in reg], anchors, (256, 256, 3), None, cfg)
        n = int(counts[i])
        assert n == ref.shape[0] == 300
        assert torch.equal(props[i, :n], ref)
    # few candidates: count < max_per_img, the tail is zero
    cfg2 = dict(cfg, nms=dict(type='nms', iou_threshold=0.05), max_per_img=2000)
    props, counts = rpn.get_bboxes_fixed(cls, reg, (256, 256, 3), cfg2, mlvl_anchors=anchors)
    for i in range(2):
        ref = rpn._get_bboxes_single([c[i] for c in cls], [r[i] for r in reg], anchors, (256, 256, 3), None, cfg2)
        n = int(counts[i])
        assert n == ref.shape[0] < 2000 and torch.equal(props[i, :n], ref) and float(props[i, n:].abs().sum()) == 0.0
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        rpn.get_bboxes_fixed(cls, reg, (256, 256, 3), cfg, mlvl_anchors=anchors)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        p2, c2 = rpn.get_bboxes_fixed(cls, reg, (256, 256, 3), cfg, mlvl_anchors=anchors)
    gr.replay()
    torch.cuda.synchronize()
    ref = rpn._get_bboxes_single([c[0] for c in cls], [r[0] for r in reg], anchors, (256, 256, 3), None, cfg)
    assert torch.equal(p2[0], ref)
