"""CPU tier: the proposal / RoI oracles against the REFERENCE'S OWN head classes run live (oracle/ref_heads.py
`load_inference`): `OrientedRPNHead._init_layers` / `forward_single` / `_get_bboxes_single` with the reference's own
`batched_nms`, `RotatedSingleRoIExtractor` with the reference's own `RoIAlignRotated` wrapper (forward and, through
autograd, backward), `RotatedShared2FCBBoxHead.forward` -- every native operator underneath is the reference's CPU C++
compiled by oracle/build_ref.py.  This pins oracle/rpn_oracle.py and oracle/roi_oracle.py, which the GPU tests
(tests/test_rpn_gpu.py, tests/test_roi_head_gpu.py) use as their checker.  Only the mmdet base classes are stand-ins
(AnchorHead: empty; BaseRoIExtractor: the constructor bookkeeping), none of their code is on these paths."""
import numpy as np
import pytest
import torch

from oracle import ref_heads as RH
from oracle import roi_oracle, rpn_oracle
from tests import synth


def _live():
    try:
        return RH.load_inference()
    except (FileNotFoundError, ImportError, OSError) as e:
        pytest.skip(f'live reference heads unavailable: {e}')


class _Cfg(dict):
    __getattr__ = dict.__getitem__

    def __deepcopy__(self, memo):
        return _Cfg({k: (_Cfg(v) if isinstance(v, dict) else v) for k, v in self.items()})


def _grid_anchors(sizes, strides):
    from sm3det_amd.rpn_head import grid_anchors
    return grid_anchors(sizes, strides, [8], [0.5, 1.0, 2.0], device='cpu')


def _reference_rpn_head(live, in_channels, feat_channels, means, stds, test_cfg):
    O = live['oriented_rpn_head']
    from oracle import ref_rpn
    _, C, _ = ref_rpn.load()
    head = O.OrientedRPNHead.__new__(O.OrientedRPNHead)
    torch.nn.Module.__init__(head)
    head.in_channels, head.feat_channels, head.num_anchors, head.cls_out_channels = in_channels, feat_channels, 3, 1
    head.use_sigmoid_cls, head.version = True, 'le90'
    head.bbox_coder = C.MidpointOffsetCoder(target_means=means, target_stds=stds, angle_range='le90')
    head.test_cfg = test_cfg
    head._init_layers()  # oriented_rpn_head.py:18-24
    return head


def test_rpn_tower_and_proposals_vs_live_reference():
    live = _live()
    torch.manual_seed(0)
    means, stds = (0.,) * 6, (1.0, 1.0, 1.0, 1.0, 0.5, 0.5)
    cfg = _Cfg(nms_pre=300, max_per_img=200, nms=_Cfg(type='nms', iou_threshold=0.8), min_bbox_size=0)
    head = _reference_rpn_head(live, 16, 16, means, stds, cfg)
    for m in (head.rpn_conv, head.rpn_cls, head.rpn_reg):
        torch.nn.init.normal_(m.weight, std=0.05)
        torch.nn.init.normal_(m.bias, std=0.05)
    sizes, strides = [(32, 32), (16, 16), (8, 8)], [4, 8, 16]
    feats = [torch.randn(1, 16, h, w) for h, w in sizes]
    params = {k: v.detach() for k, v in head.state_dict().items()}
    cls, reg = [], []
    for f in feats:
        c_ref, r_ref = head.forward_single(f)  # rotated_rpn_head.py:43-50
        c_or, r_or = rpn_oracle.rpn_forward_single(f, params)
        assert torch.equal(c_ref, c_or) and torch.equal(r_ref, r_or)
        cls.append(c_ref[0].detach())
        reg.append((r_ref[0] * 0.3).detach())
    anchors = _grid_anchors(sizes, strides)
    ref = head._get_bboxes_single(cls, reg, anchors, None, None, cfg)  # oriented_rpn_head.py:189-281
    mine = rpn_oracle.get_bboxes_single(cls, reg, anchors, dict(cfg, nms=dict(cfg['nms'])), means, stds)
    assert ref.shape == mine.shape and ref.shape[0] > 20 and ref.shape[1] == 6
    assert torch.equal(ref, mine), float((ref - mine).abs().max())
    assert bool((ref[:, 5][:-1] >= ref[:, 5][1:]).all())  # batched_nms returns score order


@pytest.mark.parametrize('aligned,clockwise', [(True, True), (False, False)])
def test_roi_extractor_forward_backward_vs_live_reference(aligned, clockwise):
    live = _live()
    strides = [4, 8, 16, 32]
    ext = live['extractor'](roi_layer=dict(type='RoIAlignRotated', out_size=7, sample_num=2, aligned=aligned,
                                           clockwise=clockwise), out_channels=8, featmap_strides=strides)
    rng = np.random.RandomState(3)
    feats = [torch.from_numpy(rng.randn(2, 8, 256 // s, 256 // s).astype(np.float32)).requires_grad_(True) for s in strides]
    rois = np.concatenate([synth.rois_for_level(40, 5 + i, batch=2, extent=256.0, wh=w)
                           for i, w in enumerate(((8., 60.), (60., 200.), (200., 700.)))])
    rois_t = torch.from_numpy(rois)
    out = ext(tuple(feats), rois_t)  # rotate_single_level_roi_extractor.py:86-140
    lv_ref = ext.map_roi_levels(rois_t, len(strides))
    mine, lv = roi_oracle.extract([f.detach() for f in feats], rois_t, strides, 7, 2, aligned, clockwise)
    assert torch.equal(lv_ref, lv) and len(set(lv.tolist())) >= 3
    assert torch.equal(out.detach(), mine), float((out.detach() - mine).abs().max())
    g = torch.from_numpy(rng.randn(*out.shape).astype(np.float32))
    out.backward(g)
    grads = roi_oracle.extract_backward(g, [tuple(f.shape) for f in feats], rois_t, strides, 7, 2, aligned, clockwise)
    for f, go in zip(feats, grads):
        ref_g = f.grad if f.grad is not None else torch.zeros_like(f)
        assert float((ref_g - go).abs().max()) <= 1e-5 * max(1.0, float(ref_g.abs().max()))


def test_shared2fc_forward_vs_live_reference():
    live = _live()
    torch.manual_seed(1)
    head = live['shared2fc'](in_channels=8, fc_out_channels=32, roi_feat_size=7, num_classes=5,
                             bbox_coder=dict(type='DeltaXYWHAOBBoxCoder', angle_range='le90', target_means=(0.,) * 5,
                                             target_stds=(0.1, 0.1, 0.2, 0.2, 0.1)),
                             reg_class_agnostic=True,
                             loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                             loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0))
    x = torch.randn(11, 8, 7, 7)
    c_ref, r_ref = head(x)  # convfc_rbbox_head.py:168-206
    c, r = roi_oracle.shared2fc_forward(x, {k: v.detach() for k, v in head.state_dict().items()})
    assert c_ref.shape == (11, 6) and r_ref.shape == (11, 5)
    assert torch.equal(c_ref, c) and torch.equal(r_ref, r)


def test_roi_head_forward_train_composition_vs_live_reference():
    """`OrientedStandardRoIHead.forward_train` of the reference (oriented_standard_roi_head.py:33-118 over
    rotate_standard_roi_head.py: assign -> sample -> rbbox2roi -> extractor -> Shared2FC -> get_targets -> loss) run
    live with the assignment of oracle/assign_oracle.py and a sampler that returns fixed picks, against the composition the
    GPU test of the product head checks itself with (tests/test_losses_gpu.py): roi_oracle.extract + shared2fc_forward +
    loss_oracle.rcnn_loss on the same picks, positives before negatives, image by image."""
    from oracle import assign_oracle
    from oracle import loss_oracle as LO
    from tests import losses_common as LC
    C = 26
    c = LC.rcnn_case(5, extent=256, P=120, num=48)
    picks = []

    class _Assigner:
        def assign(self, proposals, gts, gt_bboxes_ignore, gt_labels):
            gi, mo, lab, _ = assign_oracle.max_iou_assign(proposals[:, :5].numpy(), gts.numpy(), True,
                                                          gt_labels=gt_labels.numpy(), **LC.RCNN_ASSIGN)
            return RH._Cfg(gt_inds=torch.from_numpy(gi), max_overlaps=torch.from_numpy(np.asarray(mo)),
                           labels=torch.from_numpy(np.asarray(lab)))

    class _Sampler:
        def sample(self, assign_result, proposals, gts, gt_labels, feats=None):
            gi = assign_result.gt_inds
            pos = (gi > 0).nonzero().squeeze(1)[:10]
            neg = (gi == 0).nonzero().squeeze(1)[:30]
            assert pos.numel() >= 3 and neg.numel() >= 10
            res = RH._FixedSampling(pos, neg, proposals[:, :5], gts, gi, gt_labels)
            res.bboxes = torch.cat([res.pos_bboxes, res.neg_bboxes])
            picks.append(res)
            return res

    Head = RH.load_roi_head(lambda cfg: _Assigner(), lambda cfg, context=None: _Sampler())
    torch.manual_seed(4)
    head = Head(
        bbox_roi_extractor=dict(type='RotatedSingleRoIExtractor',
                                roi_layer=dict(type='RoIAlignRotated', out_size=7, sample_num=2, clockwise=True),
                                out_channels=8, featmap_strides=[4, 8, 16, 32]),
        bbox_head=dict(type='RotatedShared2FCBBoxHead', in_channels=8, fc_out_channels=32, roi_feat_size=7,
                       num_classes=C, reg_class_agnostic=True,
                       bbox_coder=dict(type='DeltaXYWHAOBBoxCoder', angle_range='le90', norm_factor=None, edge_swap=True,
                                       proj_xy=True, target_means=LC.RCNN_MEANS, target_stds=LC.RCNN_STDS),
                       loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                       loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0)),
        train_cfg=RH._Cfg(assigner=None, sampler=None, pos_weight=-1), version='le90')
    g = torch.Generator().manual_seed(9)
    feats = [torch.randn(2, 8, 64 >> i, 64 >> i, generator=g) for i in range(4)]
    props = []
    for i in range(2):  # proposals: jittered copies of the gts + far-away boxes
        src = c['gts'][i][torch.randint(0, c['gts'][i].shape[0], (120,), generator=g)]
        p = src + torch.randn(120, 5, generator=g) * torch.tensor([5.0, 5.0, 3.0, 2.0, 0.1])
        p[60:, :2] = torch.rand(60, 2, generator=g) * 256
        p[:, 2:4] = p[:, 2:4].clamp(min=4.0)
        props.append(torch.cat([p, torch.rand(120, 1, generator=g)], 1))
    ref = head.forward_train(feats, [dict(), dict()], props, c['gts'], c['labels'])
    assert set(ref) == {'loss_cls', 'loss_bbox', 'acc'} and len(picks) == 2
    rois = torch.cat([torch.cat([torch.full((r.bboxes.shape[0], 1), float(i)), r.bboxes], 1) for i, r in enumerate(picks)])
    x, _ = roi_oracle.extract(feats, rois, [4, 8, 16, 32], 7, 2, True, True)
    cs, bp = roi_oracle.shared2fc_forward(x, {k: v.detach() for k, v in head.bbox_head.state_dict().items()})
    exp = LO.rcnn_loss(cs, bp, [r.pos_bboxes for r in picks], [r.neg_bboxes for r in picks],
                       [r.pos_gt_bboxes for r in picks], [r.pos_gt_labels for r in picks], C, LC.RCNN_MEANS, LC.RCNN_STDS)
    for k in ('loss_cls', 'loss_bbox', 'acc'):
        torch.testing.assert_close(ref[k].detach(), exp[k].detach(), rtol=1e-6, atol=1e-7)
    assert float(ref['loss_bbox']) > 0


@pytest.mark.parametrize('rotated,cfg', [(False, dict(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, match_low_quality=True)),
                                         (True, dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=True)),
                                         (True, dict(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=False))])
def test_max_iou_rule_vs_the_copy_in_the_reference_tree(rotated, cfg):
    """oracle/assign_oracle.py's MaxIoU rule against the reference tree's own code for it:
    `MaxConvexIoUAssigner.assign_wrt_overlaps` (mmrotate/core/bbox/assigners/max_convex_iou_assigner.py:124-207), mmrotate's
    copy of mmdet's `MaxIoUAssigner.assign_wrt_overlaps` (the class the SM3Det configs name is mmdet's and absent from the
    tree).  That copy always applies the low-quality step; `match_low_quality=False` is reproduced with an unreachable
    `min_pos_iou`.  The overlaps are the oracle's (rotated: the pinned C IoU)."""
    from oracle import assign_oracle
    try:
        Rule = RH.load_assign_rule()
    except (FileNotFoundError, ImportError) as e:
        pytest.skip(str(e))
    rng = np.random.RandomState(11)
    for n, k in ((600, 7), (50, 1), (300, 12), (40, 0)):
        if rotated:
            gts = synth.rotated_boxes(k, 21 + k, extent=256.0) if k else np.zeros((0, 5), np.float32)
            src = gts[rng.randint(0, max(k, 1), n)] if k else synth.rotated_boxes(n, 5, extent=256.0)
            boxes = (src + rng.randn(n, 5).astype(np.float32) * np.array([6, 6, 4, 3, 0.15], np.float32)).astype(np.float32)
            boxes[:, 2:4] = np.maximum(boxes[:, 2:4], 2.0)
            boxes[: n // 3] = synth.rotated_boxes(n // 3, 9, extent=256.0)
            boxes[-5:] = gts[:1] if k else boxes[-5:]  # exact ties with a gt's best IoU
            ov = assign_oracle.rbbox_overlaps(gts, boxes) if k else np.zeros((0, n), np.float32)
        else:
            gts = synth.hboxes(k, 31 + k, extent=256.0) if k else np.zeros((0, 4), np.float32)
            src = gts[rng.randint(0, max(k, 1), n)] if k else synth.hboxes(n, 6, extent=256.0)
            boxes = (src + rng.randn(n, 4).astype(np.float32) * 5).astype(np.float32)
            boxes[:, 2:] = np.maximum(boxes[:, 2:], boxes[:, :2] + 2)
            boxes[: n // 3] = synth.hboxes(n // 3, 8, extent=256.0)
            boxes[-5:] = gts[:1] if k else boxes[-5:]
            ov = assign_oracle.bbox_overlaps(gts, boxes) if k else np.zeros((0, n), np.float32)
        labels = rng.randint(0, 26, k).astype(np.int64)
        rule = Rule(pos_iou_thr=cfg['pos_iou_thr'], neg_iou_thr=cfg['neg_iou_thr'],
                    min_pos_iou=cfg['min_pos_iou'] if cfg['match_low_quality'] else 2.0, gt_max_assign_all=True)
        ref = rule.assign_wrt_overlaps(torch.from_numpy(np.ascontiguousarray(ov)), torch.from_numpy(labels))
        gi, mo, lab, _ = assign_oracle.max_iou_assign(boxes, gts, rotated, gt_labels=labels, **cfg)
        assert np.array_equal(gi, ref.gt_inds.numpy()), (n, k)
        assert np.array_equal(np.asarray(lab), ref.labels.numpy()), (n, k)
        if k:
            assert np.array_equal(np.asarray(mo), ref.max_overlaps.numpy())
            assert (gi > 0).any() and (gi == 0).any()


@pytest.mark.parametrize('num,frac,ub,add_gt', [(64, 0.25, -1, True), (256, 0.5, -1, False), (64, 0.25, 3, True),
                                                (512, 0.25, -1, True)])
def test_sampler_counts_vs_live_reference_sampler(num, frac, ub, add_gt):
    """The rule of the product sampler's sync-free core (sm3det_amd/assign.py `sample_fixed_host`: plain torch, runs on CPU;
    the device kernels behind `sample_fixed` reproduce it slot for slot on the same keys, tests/test_assign_gpu.py) against the
    reference's own `RRandomSampler.sample` (rotate_random_sampler.py) run live: both draw at random, so what is compared
    is what the reference's code DETERMINES -- how many positives and negatives, that positives come first and from the
    positive set, no duplicates, gts prepended when `add_gt_as_proposals` -- over assignments with too few, exactly enough
    and too many candidates of either kind."""
    from sm3det_amd.assign import RRandomSampler as Mine
    try:
        Ref = RH.load_sampler()
    except (FileNotFoundError, ImportError) as e:
        pytest.skip(str(e))

    class _Assign:  # [memory] mmdet AssignResult.add_gt_: the gts become proposals 0..k-1 assigned to themselves
        def __init__(self, gt_inds, k):
            self.gt_inds, self.num_gts = gt_inds, k
            self.max_overlaps = torch.zeros(gt_inds.numel())
            self.labels = torch.zeros_like(gt_inds)

        def add_gt_(self, gt_labels):
            self_inds = torch.arange(1, len(gt_labels) + 1, dtype=torch.long)
            self.gt_inds = torch.cat([self_inds, self.gt_inds])
            self.max_overlaps = torch.cat([self.max_overlaps.new_ones(len(gt_labels)), self.max_overlaps])
            self.labels = torch.cat([gt_labels, self.labels])

    rng = np.random.RandomState(num + int(frac * 100) + ub)
    ref_s = Ref(num=num, pos_fraction=frac, neg_pos_ub=ub, add_gt_as_proposals=add_gt)
    mine_s = Mine(num=num, pos_fraction=frac, neg_pos_ub=ub, add_gt_as_proposals=add_gt)
    for n, p_pos, p_neg, k in ((2000, 0.05, 0.9, 8), (2000, 0.001, 0.99, 3), (300, 0.5, 0.1, 6), (40, 0.1, 0.3, 2),
                               (500, 0.0, 1.0, 4)):
        u = rng.rand(n)
        gi = np.where(u < p_pos, rng.randint(1, k + 1, n), np.where(u < p_pos + p_neg, 0, -1)).astype(np.int64)
        boxes, gts = torch.rand(n, 5), torch.rand(k, 5)
        labels = torch.randint(0, 26, (k,))
        ra = _Assign(torch.from_numpy(gi.copy()), k)
        ref = ref_s.sample(ra, boxes, gts, labels)
        full = ra.gt_inds  # after add_gt_ when applicable
        assert full.numel() == n + (k if add_gt else 0) and ref.bboxes.shape[0] == full.numel()
        idx, is_pos, valid, n_pos, n_neg = mine_s.sample_fixed_host(full)
        assert int(n_pos) == ref.pos_inds.numel() and int(n_neg) == ref.neg_inds.numel(), (n, k)
        sel_pos, sel_neg = idx[is_pos & valid], idx[(~is_pos) & valid]
        assert sel_pos.numel() == int(n_pos) and sel_neg.numel() == int(n_neg)
        assert bool((full[sel_pos] > 0).all()) and bool((full[sel_neg] == 0).all())
        assert sel_pos.unique().numel() == sel_pos.numel() and sel_neg.unique().numel() == sel_neg.numel()
        assert bool(valid[:int(n_pos) + int(n_neg)].all()) and not bool(valid[int(n_pos) + int(n_neg):].any())
        assert bool(is_pos[:int(n_pos)].all())  # positives first


# ------------------------------------------------------------------------------------------------ test-time path (g2)
def _test_path():
    try:
        return RH.load_test_path()
    except (FileNotFoundError, ImportError, OSError) as e:
        pytest.skip(f'live reference heads unavailable: {e}')


@pytest.mark.parametrize('n,nc,seed', [(400, 5, 0), (1500, 26, 1), (60, 3, 2)])
def test_multiclass_nms_rotated_vs_live_reference(n, nc, seed):
    """`multiclass_nms_rotated` of the reference (bbox_nms_rotated.py:6-96) over the reference's compiled `nms_rotated`,
    against oracle/roi_oracle.py: detections, labels and the kept candidate indices identical (incl. score filter,
    class offsets, top max_num, and the nothing-passes case)."""
    live = _test_path()
    g = torch.Generator().manual_seed(seed)
    boxes = torch.from_numpy(synth.rotated_boxes(n, seed, cluster=True))
    scores = torch.softmax(torch.randn(n, nc + 1, generator=g) * 2.0, -1)
    for thr, max_num in ((0.05, 2000), (0.2, 37), (0.999, 100)):
        d_ref, l_ref, k_ref = live['multiclass_nms_rotated'](boxes, scores, thr, _Cfg(iou_thr=0.1), max_num, return_inds=True)
        d, l, k = roi_oracle.multiclass_nms_rotated(boxes.numpy(), scores.numpy(), thr, 0.1, max_num)
        assert np.array_equal(d_ref.numpy(), d) and np.array_equal(l_ref.numpy(), l)
        if d.shape[0]:
            assert np.array_equal(k_ref.numpy(), k)
    assert d.shape[0] == 0  # the last threshold filters everything: the empty branch (:61-66)


@pytest.mark.parametrize('rescale', [False, True])
def test_roi_head_simple_test_vs_live_reference(rescale):
    """`OrientedStandardRoIHead.simple_test` of the reference run live -- `simple_test_bboxes`
    (oriented_standard_roi_head.py:126-188), `RotatedBBoxHead.get_bboxes` (rotated_bbox_head.py:358-430: softmax, the
    reference's `DeltaXYWHAOBBoxCoder.decode`, rescale, `multiclass_nms_rotated`), `rbbox2result` -- against the
    composition oracle/roi_oracle.py::simple_test the GPU test of the product head uses: per image and class the same
    number of detections, boxes and scores equal."""
    live = _test_path()
    C, strides = 6, [4, 8, 16, 32]
    means, stds = (0.,) * 5, (0.1, 0.1, 0.2, 0.2, 0.1)
    cfg = _Cfg(nms_pre=2000, min_bbox_size=0, score_thr=0.05, nms=_Cfg(iou_thr=0.1), max_per_img=2000)
    torch.manual_seed(7)
    head = live['roi_head'](
        bbox_roi_extractor=dict(type='RotatedSingleRoIExtractor',
                                roi_layer=dict(type='RoIAlignRotated', out_size=7, sample_num=2, clockwise=True),
                                out_channels=8, featmap_strides=strides),
        bbox_head=dict(type='RotatedShared2FCBBoxHead', in_channels=8, fc_out_channels=32, roi_feat_size=7,
                       num_classes=C, reg_class_agnostic=True,
                       bbox_coder=dict(type='DeltaXYWHAOBBoxCoder', angle_range='le90', norm_factor=None, edge_swap=True,
                                       proj_xy=True, target_means=means, target_stds=stds),
                       loss_cls=dict(type='CrossEntropyLoss', use_sigmoid=False, loss_weight=1.0),
                       loss_bbox=dict(type='SmoothL1Loss', beta=1.0, loss_weight=1.0)),
        test_cfg=cfg, version='le90')
    with torch.no_grad():  # spread the class scores so that several classes pass the score threshold
        head.bbox_head.fc_cls.weight.mul_(8.0)
        head.bbox_head.fc_reg.weight.mul_(10.0)
    g = torch.Generator().manual_seed(3)
    feats = [torch.randn(2, 8, 64 >> i, 64 >> i, generator=g) for i in range(4)]
    props = []
    for i in range(2):  # proposals inside the image, overlapping in clusters, sizes over all four pyramid levels
        ctr = torch.rand(30, 2, generator=g) * 200 + 28
        c = ctr[torch.randint(0, 30, (150,), generator=g)] + torch.randn(150, 2, generator=g) * 6
        w = torch.exp(torch.rand(150, generator=g) * 3.0 + 2.2)
        props.append(torch.stack([c[:, 0], c[:, 1], w, w * (0.3 + 0.6 * torch.rand(150, generator=g)),
                                  (torch.rand(150, generator=g) - 0.5) * 3.0, torch.rand(150, generator=g)], 1))
    metas = [dict(img_shape=(256, 256, 3), scale_factor=np.array([1.25, 1.25, 1.25, 1.25], np.float32)) for _ in range(2)]
    with torch.no_grad():
        ref = head.simple_test(feats, props, metas, rescale=rescale)
        raw = head.simple_test_bboxes(feats, metas, props, None)[1]
    # distinct scores: with ties the kept set would depend on the (unspecified) tie order of the sort inside NMS
    for sc in raw:
        v = sc[:, :-1].reshape(-1)
        v = v[v > 0.05]
        assert torch.unique(v).numel() == v.numel()
    params = {k: v.detach() for k, v in head.bbox_head.state_dict().items()}
    got = roi_oracle.simple_test(feats, props, metas, params, strides, C, cfg, means, stds,
                                 dict(edge_swap=True, proj_xy=True), rescale=rescale)
    assert len(ref) == len(got) == 2
    total = 0
    for r_img, g_img in zip(ref, got):
        assert len(r_img) == len(g_img) == C
        for r, q in zip(r_img, g_img):
            assert r.shape == q.shape and r.dtype == np.float32
            np.testing.assert_allclose(r, q, rtol=1e-6, atol=1e-6)
            total += r.shape[0]
    assert total > 20  # a real test: detections exist, in more than one class
    assert sum(1 for r in ref[0] if r.shape[0]) >= 2
