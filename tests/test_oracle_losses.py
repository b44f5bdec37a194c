"""CPU tier: oracle/loss_oracle.py (the restatement the GPU tests compare the HIP loss kernels with) against the
REFERENCE'S OWN head code imported live from /root/reference (oracle/ref_heads.py): OrientedRPNHead.loss ->
RotatedRPNHead.get_targets -> OrientedRPNHead._get_targets_single / loss_single, and RotatedBBoxHead.get_targets / loss.
Values and gradients.  The mmdet functions both sides share (losses, unmap, ...) are restated and unpinned -- stated in
the oracle's header.  Skipped where /root/reference is absent (GPU box)."""
import pytest
import torch

from oracle import loss_oracle as LO
from tests import losses_common as LC


def _ref():
    from oracle import ref_heads
    if not ref_heads.available():
        pytest.skip('/root/reference not present')
    return ref_heads


@pytest.mark.parametrize('seed', [0, 1])
def test_rpn_loss_oracle_equals_live_reference_head(seed):
    RH = _ref()
    c = LC.rpn_case(seed)
    cls_r = [t.clone().requires_grad_(True) for t in c['cls']]
    reg_r = [t.clone().requires_grad_(True) for t in c['reg']]
    head = RH.make_reference_rpn_head(c['anchors'], LC.RPN_ASSIGN, c['picks'], LC.RPN_MEANS, LC.RPN_STDS, 1.0 / 9.0)
    head._oracle_state['inside'] = c['inside']
    metas = [dict(img_shape=(c['extent'], c['extent'], 3)) for _ in c['gts']]
    ref = head.loss(cls_r, reg_r, c['gts'], metas)
    assert len(ref['loss_rpn_cls']) == len(c['sizes'])
    (sum(ref['loss_rpn_cls']) + 2.0 * sum(ref['loss_rpn_bbox'])).backward()

    cls_o = [t.clone().requires_grad_(True) for t in c['cls']]
    reg_o = [t.clone().requires_grad_(True) for t in c['reg']]
    oc, ob = LO.rpn_loss(cls_o, reg_o, c['anchors'], c['inside'], c['gts'], [p for p, _ in c['picks']],
                         [n for _, n in c['picks']], LC.RPN_MEANS, LC.RPN_STDS, beta=1.0 / 9.0, assign_cfg=LC.RPN_ASSIGN)
    (sum(oc) + 2.0 * sum(ob)).backward()
    for a, b in zip(oc + ob, ref['loss_rpn_cls'] + ref['loss_rpn_bbox']):
        torch.testing.assert_close(a, b, rtol=1e-6, atol=1e-7)
    assert float(sum(ob)) > 0  # positives exist: the regression branch is exercised
    for a, b in zip(cls_o + reg_o, cls_r + reg_r):
        torch.testing.assert_close(a.grad, b.grad, rtol=1e-6, atol=1e-9)


def test_rcnn_loss_oracle_equals_live_reference_head():
    RH = _ref()
    _, _, Bm = RH.load()
    c = LC.rcnn_case(3)
    head = Bm.RotatedBBoxHead(
        with_avg_pool=False, roi_feat_size=1, in_channels=8, num_classes=c['C'],
        bbox_coder=dict(type='DeltaXYWHAOBBoxCoder', angle_range='le90', norm_factor=None, edge_swap=True, proj_xy=True,
                        target_means=LC.RCNN_MEANS, target_stds=LC.RCNN_STDS),
        reg_class_agnostic=True, loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
        loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0))
    res = [RH._Cfg(pos_bboxes=pb, neg_bboxes=nb, pos_gt_bboxes=pg, pos_gt_labels=pl)
           for pb, nb, pg, pl in zip(c['pos_bboxes'], c['neg_bboxes'], c['pos_gt_bboxes'], c['pos_gt_labels'])]
    targets = head.get_targets(res, c['gts'], c['labels'], RH._Cfg(pos_weight=-1))
    cs, bp = c['cls_score'].clone().requires_grad_(True), c['bbox_pred'].clone().requires_grad_(True)
    ref = head.loss(cs, bp, None, *targets)
    (ref['loss_cls'] + 3.0 * ref['loss_bbox']).backward()

    co, bo = c['cls_score'].clone().requires_grad_(True), c['bbox_pred'].clone().requires_grad_(True)
    got = LO.rcnn_loss(co, bo, c['pos_bboxes'], c['neg_bboxes'], c['pos_gt_bboxes'], c['pos_gt_labels'], c['C'],
                       LC.RCNN_MEANS, LC.RCNN_STDS)
    (got['loss_cls'] + 3.0 * got['loss_bbox']).backward()
    for k in ('loss_cls', 'loss_bbox', 'acc'):
        torch.testing.assert_close(got[k], ref[k], rtol=1e-6, atol=1e-7)
    assert float(got['loss_bbox']) > 0
    torch.testing.assert_close(co.grad, cs.grad, rtol=1e-6, atol=1e-9)
    torch.testing.assert_close(bo.grad, bp.grad, rtol=1e-6, atol=1e-9)
