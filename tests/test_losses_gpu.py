"""GPU tier: targets + losses of the two-stage branch (sm3det_amd/det_losses.py, the `loss` / `forward_train` methods of
OrientedRPNHead, RotatedShared2FCBBoxHead, OrientedStandardRoIHead) against oracle/loss_oracle.py, which
tests/test_oracle_losses.py pins to the reference's own head classes.  The sampler is random, so each test lets the
device path sample, then hands the oracle exactly those positives / negatives (after checking them against the oracle's
own assignment)."""
import numpy as np
import pytest
import torch

from oracle import assign_oracle, loss_oracle as LO, rpn_oracle
from tests import losses_common as LC

pytestmark = pytest.mark.gpu


def _rpn_head(c, num):
    from sm3det_amd.rpn_head import OrientedRPNHead
    return OrientedRPNHead(
        in_channels=128, feat_channels=128, version='le90',
        anchor_generator=dict(type='AnchorGenerator', scales=[c['scale']], ratios=[0.5, 1.0, 2.0], strides=list(c['strides'])),
        bbox_coder=dict(type='MidpointOffsetCoder', angle_range='le90', target_means=list(LC.RPN_MEANS),
                        target_stds=list(LC.RPN_STDS)),
        loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=True, loss_weight=1.0),
        loss_bbox=dict(type='SmoothL1Loss', beta=1.0 / 9.0, loss_weight=1.0),
        train_cfg=dict(assigner=dict(type='MaxIoUAssigner', ignore_iof_thr=-1, **LC.RPN_ASSIGN),
                       sampler=dict(type='RandomSampler', num=num, pos_fraction=0.5, neg_pos_ub=-1,
                                    add_gt_as_proposals=False), allowed_border=0, pos_weight=-1, debug=False)).cuda()


@pytest.mark.parametrize('seed,nhwc', [(0, True), (1, False)])
def test_rpn_loss_and_gradients_vs_oracle(seed, nhwc):
    c = LC.rpn_case(seed)
    head = _rpn_head(c, c['num'])
    fmt = torch.channels_last if nhwc else torch.contiguous_format
    cls = [t.cuda().contiguous(memory_format=fmt).requires_grad_(True) for t in c['cls']]
    reg = [t.cuda().contiguous(memory_format=fmt).requires_grad_(True) for t in c['reg']]
    metas = [dict(img_shape=(c['extent'], c['extent'], 3)) for _ in c['gts']]
    losses, smp = head.loss(cls, reg, [g.cuda() for g in c['gts']], metas, return_samples=True)
    # anchors / inside flags / assignment of every anchor equal the oracle's
    assert torch.equal(smp['anchors'].cpu(), c['flat'])
    assert torch.equal(smp['inside'].cpu().bool(), c['inside'])
    for i, gi in enumerate(c['gt_inds']):
        assert torch.equal(smp['gt_inds'][i].cpu(), gi), i
    idx, is_pos, valid = smp['idx'].cpu(), smp['is_pos'].cpu(), smp['valid'].cpu()
    picks = [(idx[i][is_pos[i] & valid[i]], idx[i][(~is_pos[i]) & valid[i]]) for i in range(len(c['gts']))]
    for i, (p, n) in enumerate(picks):
        assert p.numel() + n.numel() == c['num'] and p.numel() <= c['num'] // 2
        assert p.numel() == len(set(p.tolist())) and n.numel() == len(set(n.tolist()))  # sampled without replacement
        assert int(smp['n_pos'][i]) == p.numel() and int(smp['n_neg'][i]) == n.numel()
    (sum(losses['loss_rpn_cls']) + 2.0 * sum(losses['loss_rpn_bbox'])).backward()

    cls_o = [t.clone().requires_grad_(True) for t in c['cls']]
    reg_o = [t.clone().requires_grad_(True) for t in c['reg']]
    oc, ob = LO.rpn_loss(cls_o, reg_o, c['anchors'], c['inside'], c['gts'], [p for p, _ in picks], [n for _, n in picks],
                         LC.RPN_MEANS, LC.RPN_STDS, beta=1.0 / 9.0, assign_cfg=LC.RPN_ASSIGN)
    (sum(oc) + 2.0 * sum(ob)).backward()
    assert float(sum(ob).detach()) > 0
    for a, b in zip(losses['loss_rpn_cls'] + losses['loss_rpn_bbox'], oc + ob):
        torch.testing.assert_close(a.detach().cpu(), b.detach(), rtol=2e-5, atol=1e-6)
    for a, b in zip(cls + reg, cls_o + reg_o):
        torch.testing.assert_close(a.grad.cpu(), b.grad, rtol=1e-4, atol=1e-7)


def test_rpn_forward_train_runs_heads_loss_and_fixed_size_proposals():
    c = LC.rpn_case(2, extent=256, strides=(4, 8, 16, 32), ks=(6, 0))  # one image without any ground truth
    head = _rpn_head(c, 64)
    head.init_weights()
    feats = [torch.randn(2, 128, h, w, device='cuda').contiguous(memory_format=torch.channels_last).requires_grad_(True)
             for h, w in c['sizes']]
    metas = [dict(img_shape=(256, 256, 3)) for _ in range(2)]
    cfg = dict(nms_pre=200, max_per_img=100, nms=dict(type='nms', iou_threshold=0.8), min_bbox_size=0)
    losses, (props, counts) = head.forward_train(feats, metas, [g.cuda() for g in c['gts']], proposal_cfg=cfg)
    assert props.shape == (2, 100, 6) and counts.shape == (2,) and int(counts.min()) > 0
    total = sum(losses['loss_rpn_cls']) + sum(losses['loss_rpn_bbox'])
    assert torch.isfinite(total)
    from sm3det_amd.rpn_head import _SplitClsReg
    hits0 = _SplitClsReg.fast_hits
    total.backward()
    assert all(f.grad is not None and torch.isfinite(f.grad).all() for f in feats)
    assert head.rpn_conv.weight.grad.abs().sum() > 0
    # the loss wrote both gradients of a level into one buffer of the fused head layout: handed on without a copy ...
    assert _SplitClsReg.fast_hits - hits0 == len(feats)
    # ... and that changes nothing: the generic path (zero map + two slice copies) gives identical gradients
    # (the sampler draws fresh random keys per call: compare the two paths on ONE graph instead)
    for t in feats + list(head.parameters()):
        t.grad = None
    losses, _ = head.forward_train(feats, metas, [g_.cuda() for g_ in c['gts']], proposal_cfg=cfg)
    total = sum(losses['loss_rpn_cls']) + sum(losses['loss_rpn_bbox'])
    leaves = feats + list(head.parameters())
    ga = torch.autograd.grad(total, leaves, retain_graph=True)
    _SplitClsReg.fast_path = False
    try:
        gb = torch.autograd.grad(total, leaves)
    finally:
        _SplitClsReg.fast_path = True
    for a, b in zip(ga, gb):  # (autograd may add the levels' contributions to a shared weight in another order: last bits)
        torch.testing.assert_close(a, b, rtol=1e-5, atol=1e-6 * float(b.abs().max()))


def _bbox_head(C):
    from sm3det_amd.roi_head import RotatedShared2FCBBoxHead
    return RotatedShared2FCBBoxHead(
        in_channels=32, fc_out_channels=64, roi_feat_size=7, num_classes=C, reg_class_agnostic=True,
        bbox_coder=dict(type='DeltaXYWHAOBBoxCoder', angle_range='le90', norm_factor=None, edge_swap=True, proj_xy=True,
                        target_means=LC.RCNN_MEANS, target_stds=LC.RCNN_STDS),
        loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
        loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0)).cuda()


def test_rcnn_loss_fused_and_reference_api_vs_oracle():
    c = LC.rcnn_case(3)
    head = _bbox_head(c['C'])
    co, bo = c['cls_score'].clone().requires_grad_(True), c['bbox_pred'].clone().requires_grad_(True)
    exp = LO.rcnn_loss(co, bo, c['pos_bboxes'], c['neg_bboxes'], c['pos_gt_bboxes'], c['pos_gt_labels'], c['C'],
                       LC.RCNN_MEANS, LC.RCNN_STDS)
    (exp['loss_cls'] + 3.0 * exp['loss_bbox']).backward()
    # (a) fixed-size block form, with two unused slots appended (valid = False) that must not change anything
    rois, gts, labels = [], [], []
    for pb, nb, pg, pl in zip(c['pos_bboxes'], c['neg_bboxes'], c['pos_gt_bboxes'], c['pos_gt_labels']):
        rois += [pb, nb]
        gts += [pg, torch.zeros(nb.shape[0], 5)]
        labels += [pl, torch.full((nb.shape[0],), c['C'], dtype=torch.long)]
    n = c['cls_score'].shape[0]
    pad = lambda t, v: torch.cat([t, t.new_full((2,) + tuple(t.shape[1:]), v)])  # noqa: E731
    cs = pad(c['cls_score'], 7.0).cuda().requires_grad_(True)
    bp = pad(c['bbox_pred'], -3.0).cuda().requires_grad_(True)
    valid = torch.cat([torch.ones(n, dtype=torch.bool), torch.zeros(2, dtype=torch.bool)]).cuda()
    got = head.loss_fused(cs, bp, pad(torch.cat(labels), 0).cuda(), valid, pad(torch.cat(rois), 1.0).cuda(),
                          pad(torch.cat(gts), 1.0).cuda())
    (got['loss_cls'] + 3.0 * got['loss_bbox']).backward()
    for k in ('loss_cls', 'loss_bbox', 'acc'):
        torch.testing.assert_close(got[k].detach().cpu(), exp[k].detach(), rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(cs.grad[:n].cpu(), co.grad, rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(bp.grad[:n].cpu(), bo.grad, rtol=1e-4, atol=1e-8)
    assert float(cs.grad[n:].abs().sum()) == 0 and float(bp.grad[n:].abs().sum()) == 0
    # (b) the reference API: get_targets(sampling_results, ...) then loss(cls_score, bbox_pred, rois, *targets)
    from types import SimpleNamespace as NS
    res = [NS(pos_bboxes=pb.cuda(), neg_bboxes=nb.cuda(), pos_gt_bboxes=pg.cuda(), pos_gt_labels=pl.cuda())
           for pb, nb, pg, pl in zip(c['pos_bboxes'], c['neg_bboxes'], c['pos_gt_bboxes'], c['pos_gt_labels'])]
    targets = head.get_targets(res, None, None, dict(pos_weight=-1))
    cs2, bp2 = c['cls_score'].cuda().requires_grad_(True), c['bbox_pred'].cuda().requires_grad_(True)
    got2 = head.loss(cs2, bp2, None, *targets)
    (got2['loss_cls'] + 3.0 * got2['loss_bbox']).backward()
    for k in ('loss_cls', 'loss_bbox', 'acc'):
        torch.testing.assert_close(got2[k].detach().cpu(), exp[k].detach(), rtol=2e-5, atol=1e-6)
    torch.testing.assert_close(cs2.grad.cpu(), co.grad, rtol=1e-4, atol=1e-8)
    torch.testing.assert_close(bp2.grad.cpu(), bo.grad, rtol=1e-4, atol=1e-8)


def test_roi_head_forward_train_samples_match_oracle_assignment_and_loss():
    from sm3det_amd.roi_head import OrientedStandardRoIHead
    C = 26
    c = LC.rcnn_case(5, extent=256, P=200, num=64)
    head = OrientedStandardRoIHead(
        bbox_roi_extractor=dict(type='RotatedSingleRoIExtractor',
                                roi_layer=dict(type='RoIAlignRotated', out_size=7, sample_num=2, clockwise=True),
                                out_channels=32, featmap_strides=[4, 8, 16, 32]),
        bbox_head=dict(type='RotatedShared2FCBBoxHead', in_channels=32, fc_out_channels=64, roi_feat_size=7,
                       num_classes=C, reg_class_agnostic=True,
                       bbox_coder=dict(type='DeltaXYWHAOBBoxCoder', angle_range='le90', norm_factor=None, edge_swap=True,
                                       proj_xy=True, target_means=LC.RCNN_MEANS, target_stds=LC.RCNN_STDS),
                       loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                       loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0)),
        train_cfg=dict(assigner=dict(type='MaxIoUAssigner', iou_calculator=dict(type='RBboxOverlaps2D'), ignore_iof_thr=-1,
                                     **LC.RCNN_ASSIGN),
                       sampler=dict(type='RRandomSampler', num=64, pos_fraction=0.25, neg_pos_ub=-1,
                                    add_gt_as_proposals=True), pos_weight=-1, debug=False), version='le90').cuda()
    head.init_weights()
    g = torch.Generator().manual_seed(9)
    feats = [torch.randn(2, 32, 64 >> i, 64 >> i, generator=g).cuda().contiguous(memory_format=torch.channels_last)
             .requires_grad_(True) for i in range(4)]
    B, P = 2, 200
    props = torch.zeros(B, P, 6)
    counts = torch.tensor([190, 170])
    for i in range(B):  # proposals: jittered copies of the gts + far-away boxes; rows past `counts` hold junk
        src = c['gts'][i][torch.randint(0, c['gts'][i].shape[0], (P,), generator=g)]
        props[i, :, :5] = src + torch.randn(P, 5, generator=g) * torch.tensor([5.0, 5.0, 3.0, 2.0, 0.1])
        props[i, P // 2:, :2] = torch.rand(P - P // 2, 2, generator=g) * 256
        props[i, :, 2:4] = props[i, :, 2:4].clamp(min=4.0)
        props[i, counts[i]:, :5] = c['gts'][i][0]  # junk rows equal to a gt: would be positives if they were not masked
    gtb, gtl = [x.cuda() for x in c['gts']], [x.cuda() for x in c['labels']]
    losses, smp = head.forward_train(feats, None, (props.cuda(), counts.cuda()), gtb, gtl, return_samples=True)
    (losses['loss_cls'] + losses['loss_bbox']).backward()
    assert all(f.grad is not None for f in feats) and head.bbox_head.fc_reg.weight.grad is not None
    rois, labels, valid, gts_s = smp['rois'].cpu(), smp['labels'].cpu(), smp['valid'].cpu(), smp['gts'].cpu()
    S = 64
    pb, nb, pg, pl, rows = [], [], [], [], []
    for i in range(B):
        k = c['gts'][i].shape[0]
        cand = torch.cat([c['gts'][i], props[i, :counts[i], :5]])  # what the reference would sample from
        gi, _, lab, _ = assign_oracle.max_iou_assign(cand[k:].numpy(), c['gts'][i].numpy(), True,
                                                     gt_labels=c['labels'][i].numpy(), **LC.RCNN_ASSIGN)
        gi = np.concatenate([np.arange(1, k + 1), gi])
        sl = slice(i * S, (i + 1) * S)
        r, l, v, gs = rois[sl], labels[sl], valid[sl], gts_s[sl]
        assert bool((r[:, 0] == i).all()) and bool(v.all())  # enough candidates: every slot used
        for j in range(S):  # each sampled box is one of the candidates, with the oracle's assignment
            d = (cand - r[j, 1:]).abs().sum(1)
            m = int(d.argmin())
            assert float(d[m]) < 1e-4
            if l[j] < C:
                assert gi[m] > 0 and int(l[j]) == int(c['labels'][i][gi[m] - 1])
                assert torch.allclose(gs[j], c['gts'][i][gi[m] - 1])
            else:
                assert gi[m] == 0
        pos = l < C
        assert int(pos.sum()) <= 16 and bool((pos[:int(pos.sum())]).all())  # positives first, at most num * fraction
        pb.append(r[pos, 1:]); nb.append(r[~pos, 1:]); pg.append(gs[pos]); pl.append(l[pos])
    with torch.no_grad():
        res = head._bbox_forward(feats, smp['rois'])
    exp = LO.rcnn_loss(res['cls_score'].cpu(), res['bbox_pred'].cpu(), pb, nb, pg, pl, C, LC.RCNN_MEANS, LC.RCNN_STDS)
    for k_ in ('loss_cls', 'loss_bbox', 'acc'):
        torch.testing.assert_close(losses[k_].detach().cpu(), exp[k_], rtol=2e-5, atol=1e-6)
