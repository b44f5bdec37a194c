"""GPU tier: MaxIoU assignment + random sampling (sm3det_amd/assign.py -> sm3_max_iou_assign) against the numpy
restatement oracle/assign_oracle.py (rule: mmdet, parity unpinned; rotated IoU: pinned C oracle).

Bar: gt_inds / labels identical (integer equality); the only tolerated differences are boxes whose IoU with some gt
lies within 1e-6 of a threshold or of the gt's best IoU (box_iou_rotated on the device matches the C oracle bit for bit
except for <= 1e-4 of the pairs, where it is 1 ulp off: tests/test_ops_gpu.py) -- counted and bounded."""
import numpy as np
import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu
dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731


def _check(got_inds, got_ov, exp_inds, exp_ov, ov, thrs):
    got_inds, got_ov = got_inds.cpu().numpy(), got_ov.cpu().numpy()
    if got_inds.size == 0:
        assert exp_inds.size == 0
        return
    assert np.abs(got_ov - exp_ov).max() <= 1e-6
    bad = np.nonzero(got_inds != exp_inds)[0]
    if ov is not None:
        for j in bad:  # every mismatch must sit on a threshold / tie within float noise
            col = ov[:, j]
            near = min(min(abs(col.max() - t) for t in thrs), np.abs(ov.max(1) - col).min())
            assert near <= 2e-6, (j, got_inds[j], exp_inds[j], near)
    assert len(bad) <= max(1, len(exp_inds) // 2000), len(bad)


@pytest.mark.parametrize('n,k,seed', [(2000, 8, 0), (2000, 64, 1), (513, 1, 2), (100, 0, 3), (0, 5, 4)])
def test_rcnn_assigner_rotated_vs_oracle(n, k, seed):
    """rcnn stage of main_SM3Det.py: MaxIoUAssigner(0.5, 0.5, 0.5, match_low_quality=False, RBboxOverlaps2D)"""
    from oracle import assign_oracle as AO
    from sm3det_amd.assign import MaxIoUAssigner
    gts = synth.rotated_boxes(k, seed + 50) if k else np.zeros((0, 5), np.float32)
    props = synth.rotated_boxes(n, seed, cluster=True) if n else np.zeros((0, 5), np.float32)
    if n and k:  # make sure positives exist: jittered copies of the gts
        rng = np.random.RandomState(seed)
        m = min(n // 4, 8 * k)
        props[:m] = gts[rng.randint(0, k, m)] + rng.normal(0, 1.5, (m, 5)).astype(np.float32) * [1, 1, 1, 1, 0.02]
    labels = np.random.RandomState(seed).randint(0, 26, k).astype(np.int64)
    for mlq in (False, True):
        a = MaxIoUAssigner(pos_iou_thr=0.5, neg_iou_thr=0.5, min_pos_iou=0.5, match_low_quality=mlq,
                           iou_calculator=dict(type='RBboxOverlaps2D'), ignore_iof_thr=-1)
        r = a.assign(dev(props), dev(gts), None, dev(labels))
        ei, eo, el, ov = AO.max_iou_assign(props, gts, True, 0.5, 0.5, 0.5, mlq, labels)
        assert r.num_gts == k and r.gt_inds.dtype == torch.long
        _check(r.gt_inds, r.max_overlaps, ei, eo, ov, (0.5,))
        same = r.gt_inds.cpu().numpy() == ei
        assert np.array_equal(r.labels.cpu().numpy()[same], el[same])
    if n and k:
        assert (r.gt_inds > 0).sum() > 0


@pytest.mark.parametrize('levels,k,seed', [((64, 32, 16, 8, 4), 8, 0), ((256, 128, 64, 32, 16), 20, 1)])
def test_rpn_assigner_horizontal_anchor_grid_vs_oracle(levels, k, seed):
    """rpn stage: MaxIoUAssigner(0.7, 0.3, 0.3, match_low_quality=True) with BboxOverlaps2D over the FULL multi-level
    anchor grid (second case = the real 1024^2 grid: 261 888 anchors) vs the hbb of rotated gts"""
    from oracle import assign_oracle as AO
    from sm3det_amd.assign import MaxIoUAssigner
    from sm3det_amd.rpn_head import grid_anchors
    strides = [4, 8, 16, 32, 64]
    anchors = torch.cat(grid_anchors([(s, s) for s in levels], strides, [8], [0.5, 1.0, 2.0], device='cuda')).contiguous()
    ext = float(levels[0] * strides[0])
    hb = synth.hboxes(k, seed + 9, extent=ext, wh=(16.0, min(256.0, ext / 2)))
    a = MaxIoUAssigner(pos_iou_thr=0.7, neg_iou_thr=0.3, min_pos_iou=0.3, match_low_quality=True, ignore_iof_thr=-1)
    r = a.assign(anchors, dev(hb), None, None)
    ei, eo, _, ov = AO.max_iou_assign(anchors.cpu().numpy(), hb, False, 0.7, 0.3, 0.3, True, None)
    assert r.labels is None
    _check(r.gt_inds, r.max_overlaps, ei, eo, ov, (0.7, 0.3))
    gi = r.gt_inds.cpu().numpy()
    assert (gi > 0).sum() >= k and (gi == 0).sum() > 0 and (gi == -1).sum() > 0  # every gt got its best anchor


@pytest.mark.parametrize('n,n_pos,n_ign,num,frac,ub', [
    (261888, 150, 9000, 256, 0.5, -1), (261888, 40, 0, 256, 0.5, -1), (2008, 700, 30, 512, 0.25, -1), (2008, 3, 0, 512, 0.25, -1),
    (300, 20, 100, 512, 0.25, -1), (5000, 64, 10, 256, 0.5, 3), (7, 0, 0, 256, 0.5, -1), (64, 64, 0, 256, 0.5, -1),
    (100000, 60000, 100, 2048, 0.5, -1)])
def test_sampler_kernels_equal_the_host_rule_on_the_same_keys(n, n_pos, n_ign, num, frac, ub):
    """sampler.hip (count / list under a key threshold / LDS sort + slots) against `sample_fixed_host` (two full stable sorts of
    the masked keys) on identical keys: the same valid slots in the same order, the same counts -- the RPN stage's 261 888
    anchors, the RCNN stage's 2000 + gts, fewer candidates than slots, no positives, only positives, neg_pos_ub."""
    from sm3det_amd.assign import RandomSampler
    g = torch.Generator(device='cuda').manual_seed(n + n_pos)
    gt_inds = torch.zeros(n, dtype=torch.long, device='cuda')
    perm = torch.randperm(n, device='cuda', generator=g)
    gt_inds[perm[:n_pos]] = torch.randint(1, 9, (n_pos,), device='cuda', generator=g)
    gt_inds[perm[n_pos:n_pos + n_ign]] = -1
    s = RandomSampler(num=num, pos_fraction=frac, neg_pos_ub=ub, add_gt_as_proposals=False)
    for rep in range(3):
        key = torch.rand(n, device='cuda', generator=g)
        if rep == 2 and n > 16:
            key[: n // 2] = key[0]  # heavy ties: broken by index in both forms
        a = s.sample_fixed(gt_inds, key=key)
        b = s.sample_fixed_host(gt_inds, key=key)
        assert int(a[3]) == int(b[3]) and int(a[4]) == int(b[4])
        v = b[2]
        assert torch.equal(a[2], v) and torch.equal(a[1], b[1])
        assert torch.equal(a[0][v], b[0][v])
        assert bool((a[0] >= 0).all()) and (n == 0 or bool((a[0] < n).all()))


def test_sampler_short_list_is_rebuilt_exactly():
    """the emit kernel's slow path: keys that are NOT uniform -- none below 0.5 -- leave the candidate list empty, and
    doubling the threshold past 0.5 overflows it; the workgroup bisects the threshold and still returns the exact smallest
    keys.  Second case: every key below the first threshold (the fast list overflows)."""
    from sm3det_amd.assign import RandomSampler
    n = 50000
    gt_inds = torch.zeros(n, dtype=torch.long, device='cuda')
    gt_inds[:20000] = 1
    key = torch.rand(n, device='cuda', generator=torch.Generator(device='cuda').manual_seed(5)) * 0.5 + 0.5  # none below 0.5
    s = RandomSampler(num=256, pos_fraction=0.5, neg_pos_ub=-1, add_gt_as_proposals=False)
    a, b = s.sample_fixed(gt_inds, key=key), s.sample_fixed_host(gt_inds, key=key)
    assert int(a[3]) == 128 and int(a[4]) == 128 and bool(a[2].all())
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    key = torch.rand(n, device='cuda') * 1e-4
    a, b = s.sample_fixed(gt_inds, key=key), s.sample_fixed_host(gt_inds, key=key)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1]) and bool(a[2].all())
    # one key value everywhere: no threshold separates 128 of 20 000 equal keys; after 64 rounds the truncated list is used --
    # still a valid sample (right classes, in range, no duplicates)
    key = torch.full((n,), 0.25, device='cuda')
    idx, is_pos, valid, n_pos, n_neg = s.sample_fixed(gt_inds, key=key)
    assert int(n_pos) == 128 and int(n_neg) == 128 and bool(valid.all())
    assert bool((gt_inds[idx[is_pos]] > 0).all()) and bool((gt_inds[idx[~is_pos]] == 0).all()) and idx.unique().numel() == 256
    # the scratch counters are left zero: the next call on the same stream is correct
    key2 = torch.rand(n, device='cuda')
    a, b = s.sample_fixed(gt_inds, key=key2), s.sample_fixed_host(gt_inds, key=key2)
    assert torch.equal(a[0], b[0])


def test_random_sampler_rule_and_uniformity():
    from sm3det_amd.assign import AssignResult, RRandomSampler, RandomSampler
    g = torch.Generator(device='cuda').manual_seed(0)
    n = 3000
    gt_inds = torch.zeros(n, dtype=torch.long, device='cuda')
    gt_inds[:400] = torch.randint(1, 9, (400,), device='cuda', generator=g)
    gt_inds[400:500] = -1
    s = RandomSampler(num=512, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=False)
    idx, is_pos, valid, n_pos, n_neg = s.sample_fixed(gt_inds, g)
    assert int(n_pos) == 128 and int(n_neg) == 384 and bool(valid.all())
    assert bool((gt_inds[idx[is_pos]] > 0).all()) and bool((gt_inds[idx[~is_pos]] == 0).all())
    assert idx.unique().numel() == 512  # without replacement
    # fewer positives than expected: negatives fill the batch (BaseSampler.sample: num - num_sampled_pos)
    gi2 = torch.zeros(n, dtype=torch.long, device='cuda')
    gi2[:10] = 1
    idx, is_pos, valid, n_pos, n_neg = s.sample_fixed(gi2, g)
    assert int(n_pos) == 10 and int(n_neg) == 502 and int(valid.sum()) == 512
    # fewer candidates than slots: the rest is flagged invalid
    gi3 = torch.full((n,), -1, dtype=torch.long, device='cuda')
    gi3[:5], gi3[5:25] = 2, 0
    idx, is_pos, valid, n_pos, n_neg = s.sample_fixed(gi3, g)
    assert int(n_pos) == 5 and int(n_neg) == 20 and int(valid.sum()) == 25
    # neg_pos_ub
    s2 = RandomSampler(num=256, pos_fraction=0.5, neg_pos_ub=3, add_gt_as_proposals=False)
    _, _, valid, n_pos, n_neg = s2.sample_fixed(gi2, g)
    assert int(n_pos) == 10 and int(n_neg) == 30
    # uniformity: each of 400 positives is picked with probability 128/400 over repeated draws
    cnt = torch.zeros(n, device='cuda')
    for _ in range(200):
        idx, is_pos, valid, _, _ = s.sample_fixed(gt_inds, g)
        cnt[idx[is_pos]] += 1
    p = cnt[:400] / 200
    assert abs(float(p.mean()) - 0.32) < 0.01 and float(p.std()) < 0.06
    # mmdet API with add_gt_as_proposals (RRandomSampler of the rcnn stage)
    gts = dev(synth.rotated_boxes(6, 3))
    props = dev(synth.rotated_boxes(1000, 4))
    ar = AssignResult(6, torch.zeros(1000, dtype=torch.long, device='cuda'), torch.zeros(1000, device='cuda'),
                      torch.full((1000,), -1, dtype=torch.long, device='cuda'))
    res = RRandomSampler(num=512, pos_fraction=0.25, neg_pos_ub=-1, add_gt_as_proposals=True).sample(
        ar, props, gts, torch.arange(6, device='cuda'), generator=g)
    assert res.pos_inds.numel() == 6 and bool(res.pos_is_gt.all()) and res.neg_inds.numel() == 506
    assert torch.equal(res.pos_gt_bboxes, gts[res.pos_assigned_gt_inds]) and res.bboxes.shape == (512, 5)
    assert torch.equal(torch.sort(res.pos_gt_labels)[0], torch.arange(6, device='cuda'))
