"""GPU tier (-m gpu): HIP operators, called through the C ABI (sm3det_amd.mmcv_ext -> libsm3det_hip.so), against
the CPU oracle (oracle/ops_oracle.c), the compiled reference (oracle/_ref, when its .so travelled) and the mmcv
golden vectors.

Tolerances (stated here as the task requires):
  * nms / nms_rotated keep lists: BIT-EXACT (int64 equality) on tie-free scores;
  * box_iou_rotated: bit-exact expected (same op order, -ffp-contract=off); asserted <= 1e-6 abs with identical
    support (>0 pattern) and the mismatch count reported -- device double cos/sin (ocml) may differ from glibc in
    the last ulp before the cast to float;
  * RoIAlignRotated fwd: <= 1e-5 rel; bwd: <= 1e-4 abs/rel (fp32 atomics => summation order differs).
"""
import os

import numpy as np
import pytest
import torch

from tests import golden_vectors as GV
from tests import synth

pytestmark = pytest.mark.gpu


def _ops():
    from sm3det_amd import mmcv_ops
    return mmcv_ops


def _oracle():
    from oracle import ops_oracle
    return ops_oracle


def _ref_or_none():
    """the compiled reference operators (oracle/_ref, built from /root/reference by oracle/build_ref.py): None only when
    the .so is ABSENT (callers skip that comparison with the reason); a .so that exists but does not load RAISES, so a
    broken build cannot turn the reference comparisons into silent no-ops."""
    from oracle import build_ref
    if not os.path.exists(build_ref.so_path()):
        return None
    return build_ref.load_ref()


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


# ----------------------------------------------------------------------------------- golden vectors
def test_box_iou_rotated_golden():
    g = GV.BOX_IOU_ROTATED
    ops = _ops()
    b1, b2 = dev(g['boxes1']), dev(g['boxes2'])
    assert np.allclose(ops.box_iou_rotated(b1, b2).cpu().numpy(), g['ious'], atol=1e-4)
    assert np.allclose(ops.box_iou_rotated(b1, b2, aligned=True).cpu().numpy(), np.diag(g['ious']), atol=1e-4)
    # ccw definition (test_box_iou_rotated.py:35-44)
    b1[..., -1] *= -1
    b2[..., -1] *= -1
    assert np.allclose(ops.box_iou_rotated(b1, b2, clockwise=False).cpu().numpy(), g['ious'], atol=1e-4)


def test_nms_rotated_golden():
    g = GV.NMS_ROTATED
    ops = _ops()
    boxes = dev(g['dets'])
    dets, keep = ops.nms_rotated(boxes[:, :5], boxes[:, -1], g['thr'])
    assert keep.cpu().tolist() == g['keep']
    assert np.allclose(dets.cpu().numpy()[:, :5], g['dets'][g['keep'], :5])
    # with labels: the reference CPU path ignores them (class-agnostic result)
    dets, keep = ops.nms_rotated(boxes[:, :5], boxes[:, -1], g['thr'], dev(g['labels']))
    assert keep.cpu().tolist() == g['keep']
    # ccw
    b = boxes.clone()
    b[..., -2] *= -1
    dets, keep = ops.nms_rotated(b[:, :5], b[:, -1], g['thr'], clockwise=False)
    assert keep.cpu().tolist() == g['keep']


def test_batched_nms_rotated_golden():
    g = GV.NMS_ROTATED
    ops = _ops()
    boxes = dev(g['dets'])
    cfg = dict(type='nms_rotated', iou_threshold=g['thr'])
    b, keep = ops.batched_nms(boxes[:, :5], boxes[:, -1], dev(g['labels']), cfg, class_agnostic=True)
    assert keep.cpu().tolist() == g['keep']
    b, keep = ops.batched_nms(boxes[:, :5], boxes[:, -1], dev(g['labels']), cfg, class_agnostic=False)
    assert keep.cpu().tolist() == g['keep_per_class']


def test_nms_golden():
    g = GV.NMS
    ops = _ops()
    dets, inds = ops.nms(dev(g['boxes']), dev(g['scores']), iou_threshold=g['thr'], offset=0)
    assert inds.cpu().tolist() == g['keep']
    assert np.allclose(dets.cpu().numpy()[:, :4], g['boxes'][g['keep']])


@pytest.mark.parametrize('case', range(len(GV.ROI_ALIGN_ROTATED_CASES)))
@pytest.mark.parametrize('channels_last', [False, True])
def test_roi_align_rotated_golden(case, channels_last):
    ops = _ops()
    x, rois, out, grad = GV.ROI_ALIGN_ROTATED_CASES[case]
    x = dev(np.array(x, np.float32)).requires_grad_(True)
    xin = x.contiguous(memory_format=torch.channels_last) if channels_last else x
    rois = dev(np.array(rois, np.float32))
    layer = ops.RoIAlignRotated(out_size=2, spatial_scale=1.0, sample_num=2)  # deprecated aliases on purpose
    assert layer.output_size == (2, 2) and layer.sampling_ratio == 2
    y = layer(xin, rois)
    y.backward(torch.ones_like(y))
    assert np.allclose(y.detach().cpu().numpy(), np.array(out, np.float32), atol=1e-3)
    assert np.allclose(x.grad.cpu().numpy(), np.array(grad, np.float32), atol=1e-3)


# ----------------------------------------------------------------------------------- randomized vs oracle
@pytest.mark.parametrize('n1,n2,seed,cluster', [(2000, 64, 0, False), (2000, 512, 1, True), (777, 33, 2, True)])
def test_box_iou_rotated_vs_oracle(n1, n2, seed, cluster):
    ops, O = _ops(), _oracle()
    b1 = synth.rotated_boxes(n1, seed, cluster=cluster)
    b2 = synth.rotated_boxes(n2, seed + 100, cluster=cluster)
    for mode, flag in (('iou', 0), ('iof', 1)):
        got = ops.box_iou_rotated(dev(b1), dev(b2), mode=mode).cpu().numpy()
        exp = O.box_iou_rotated(b1, b2, flag)
        diff = np.abs(got - exp)
        nmis = int((got != exp).sum())
        print(f'box_iou_rotated {n1}x{n2} {mode}: max|d|={diff.max():.3g}, non-bit-exact={nmis}/{got.size}')
        assert diff.max() <= 1e-6
        assert np.array_equal(got > 0, exp > 0)
        assert nmis <= got.size * 1e-4


def test_box_iou_rotated_degenerate_and_empty():
    ops, O = _ops(), _oracle()
    b1, b2 = synth.degenerate_rotated_pairs()
    got = ops.box_iou_rotated(dev(b1), dev(b2), aligned=True).cpu().numpy()
    assert np.array_equal(got, O.box_iou_rotated(b1, b2, 0, True))
    got = ops.box_iou_rotated(dev(b1), dev(b2)).cpu().numpy()
    assert np.array_equal(got, O.box_iou_rotated(b1, b2, 0, False))
    e = ops.box_iou_rotated(torch.zeros(0, 5).cuda(), dev(b2))
    assert e.shape == (0, len(b2))


def test_box_iou_rotated_vs_compiled_reference():
    ref = _ref_or_none()
    if ref is None:
        pytest.skip('oracle/_ref .so did not travel')
    ops = _ops()
    b1 = synth.rotated_boxes(500, 11, cluster=True)
    b2 = synth.rotated_boxes(100, 12, cluster=True)
    out = torch.zeros(500 * 100)
    ref.box_iou_rotated(torch.from_numpy(b1), torch.from_numpy(b2), out, 0, False)
    got = ops.box_iou_rotated(dev(b1), dev(b2)).cpu().numpy().reshape(-1)
    assert np.abs(got - out.numpy()).max() <= 1e-6


@pytest.mark.parametrize('n,thr,cluster', [(2000, 0.1, False), (2000, 0.1, True), (1000, 0.5, True), (1, 0.3, False),
                                           (65, 0.1, True), (4097, 0.1, False)])
def test_nms_rotated_vs_oracle(n, thr, cluster):
    ops, O = _ops(), _oracle()
    d = synth.rotated_boxes(n, 3, cluster=cluster)
    s = synth.unique_scores(n, 4)
    dets, keep = ops.nms_rotated(dev(d), dev(s), thr)
    exp = O.nms_rotated(d, s, thr)
    assert keep.dtype == torch.int64
    assert np.array_equal(keep.cpu().numpy(), exp), (len(exp), keep.numel())


@pytest.mark.parametrize('n,thr,offset,cluster', [(8768, 0.8, 0, True), (5000, 0.6, 1, True), (300, 0.3, 0, False),
                                                  (64, 0.5, 0, True), (20000, 0.7, 0, True)])
def test_nms_vs_oracle(n, thr, offset, cluster):
    ops, O = _ops(), _oracle()
    b = synth.hboxes(n, 5, cluster=cluster)
    s = synth.unique_scores(n, 6)
    dets, keep = ops.nms(dev(b), dev(s), iou_threshold=thr, offset=offset)
    exp = O.nms(b, s, thr, offset)
    assert np.array_equal(keep.cpu().numpy(), exp), (len(exp), keep.numel())


@pytest.mark.parametrize('n,step', [(64, 6.0), (200, 6.0), (1000, 6.5), (130, 2.0)])
def test_nms_chains_of_dependent_boxes_vs_oracle(n, step):
    """Longest possible suppression chains: a row of 10-px boxes shifted by `step` px in score order -- box i overlaps
    i + 1 (and, for the small step, several more) above the threshold, so whether box i survives depends on the fate of
    every earlier box: the greedy order inside a 64-box block then needs as many rounds of the sweep's parallel fixed
    point as the chain is long (ops_rotated.hip nms_sweep_lds_kernel), and the chains cross block boundaries."""
    ops, O = _ops(), _oracle()
    x = (np.arange(n) * step).astype(np.float32)
    hb = np.stack([x, np.zeros(n, np.float32), x + 10, np.full(n, 10, np.float32)], 1)
    s = np.linspace(1.0, 0.1, n).astype(np.float32)  # strictly decreasing: score order = position order
    _, keep = ops.nms(dev(hb), dev(s), iou_threshold=0.2, offset=0)
    exp = O.nms(hb, s, 0.2, 0)
    assert np.array_equal(keep.cpu().numpy(), exp), (len(exp), keep.numel())
    assert 1 < len(exp) < n
    rb = np.stack([x + 5, np.full(n, 5, np.float32), np.full(n, 10, np.float32), np.full(n, 10, np.float32),
                   np.full(n, 0.02, np.float32)], 1)
    _, keep = ops.nms_rotated(dev(rb), dev(s), 0.2)
    exp = O.nms_rotated(rb, s, 0.2)
    assert np.array_equal(keep.cpu().numpy(), exp), (len(exp), keep.numel())
    assert 1 < len(exp) < n


def test_nms_score_threshold_max_num_numpy_and_empty():
    ops, O = _ops(), _oracle()
    b = synth.hboxes(500, 8, cluster=True)
    s = synth.unique_scores(500, 9)
    dets, keep = ops.nms(b, s, iou_threshold=0.5, score_threshold=0.3, max_num=20)  # numpy in -> numpy out
    assert isinstance(keep, np.ndarray) and len(keep) == 20
    valid = np.nonzero(s > 0.3)[0]
    exp = valid[O.nms(b[valid], s[valid], 0.5, 0)][:20]
    assert np.array_equal(keep, exp)
    d, k = ops.nms(torch.zeros(0, 4).cuda(), torch.zeros(0).cuda(), iou_threshold=0.5)
    assert k.numel() == 0 and d.shape == (0, 5)
    d, k = ops.nms_rotated(torch.zeros(0, 5).cuda(), torch.zeros(0).cuda(), 0.5)
    assert k is None


def test_nms_idempotent_property():
    """size-independent property: NMS of the kept set keeps everything (all pairwise IoU <= thr)."""
    ops = _ops()
    n = 8768
    b = dev(synth.hboxes(n, 21, cluster=True))
    s = dev(synth.unique_scores(n, 22))
    _, keep = ops.nms(b, s, iou_threshold=0.8)
    _, keep2 = ops.nms(b[keep], s[keep], iou_threshold=0.8)
    assert keep2.numel() == keep.numel() and torch.equal(keep2, torch.arange(keep.numel(), device='cuda'))
    d = dev(synth.rotated_boxes(10000, 23, cluster=True))
    s = dev(synth.unique_scores(10000, 24))
    _, k = ops.nms_rotated(d, s, 0.1)
    _, k2 = ops.nms_rotated(d[k], s[k].contiguous(), 0.1)
    assert k2.numel() == k.numel()
    # scores of kept boxes are descending
    assert torch.all(s[k][:-1] >= s[k][1:])
    with pytest.raises(RuntimeError):  # shape mismatch must raise, not read out of bounds
        ops.nms_rotated(d, s[:100].contiguous(), 0.1)


def test_argsort_matches_stable_sort():
    from sm3det_amd import _lib
    L = _lib.lib()
    for n in (1, 5, 4096, 4097, 10000, 70000):
        s = torch.from_numpy(np.random.RandomState(n).randint(0, 50, size=n).astype(np.float32)).cuda()  # many ties
        order = torch.empty(n, dtype=torch.long, device='cuda')
        nb = L.sm3_argsort_desc_workspace_bytes(n)
        ws = _lib.workspace(nb, s.device)
        _lib.check(L.sm3_argsort_desc_f32(_lib.ptr(s), n, _lib.ptr(order), _lib.ptr(ws), nb, _lib.stream_ptr()), 'sort')
        exp = torch.sort(s, descending=True, stable=True)[1]
        assert torch.equal(order, exp), n


def test_topk_equals_prefix_of_stable_descending_sort():
    """merge-tree top-k (k <= 2048) == the first k entries of the stable descending sort, on tie-heavy and tie-free
    scores, chunk counts that are odd / even / one, k at and below the run length; larger k is refused"""
    from sm3det_amd import _lib
    L = _lib.lib()
    for n, k, ties in ((1, 1, True), (100, 17, True), (4096, 2000, True), (4097, 2048, False), (12295, 2000, True),
                       (49152, 2000, False), (196608, 2000, True), (196608, 2048, False), (70001, 1, True)):
        rs = np.random.RandomState(n + k)
        v = rs.randint(0, 500, size=n).astype(np.float32) if ties else rs.permutation(n).astype(np.float32)
        sc = torch.from_numpy(v).cuda()
        order = torch.full((k,), -1, dtype=torch.long, device='cuda')
        nb = L.sm3_topk_desc_workspace_bytes(n)
        ws = _lib.workspace(nb, sc.device)
        _lib.check(L.sm3_topk_desc_f32(_lib.ptr(sc), n, k, _lib.ptr(order), _lib.ptr(ws), nb, _lib.stream_ptr()), 'topk')
        exp = torch.sort(sc, descending=True, stable=True)[1][:k]
        assert torch.equal(order, exp), (n, k)
    rc = L.sm3_topk_desc_f32(_lib.ptr(sc), n, 4096, _lib.ptr(order), _lib.ptr(ws), nb, _lib.stream_ptr())
    assert rc != 0  # SM3_ERR_UNSUPPORTED: the caller falls back to the argsort


@pytest.mark.parametrize('aligned,clockwise,ratio', [(True, True, 2), (True, False, 0), (False, True, 2)])
@pytest.mark.parametrize('channels_last', [False, True])
def test_roi_align_rotated_vs_oracle(aligned, clockwise, ratio, channels_last):
    ops, O = _ops(), _oracle()
    rng = np.random.RandomState(0)
    x = rng.randn(2, 16, 40, 48).astype(np.float32)
    rois = synth.rois_for_level(80, 1, batch=2, extent=48 * 4.0, wh=(4.0, 120.0))
    rois[0, 1:3] = [-20, -20]
    rois[1, 1:3] = [400, 300]
    rois[2, 3:5] = [0.5, 0.5]
    xt = dev(x).requires_grad_(True)
    xin = xt.contiguous(memory_format=torch.channels_last) if channels_last else xt
    y = ops.roi_align_rotated(xin, dev(rois), 7, 0.25, ratio, aligned, clockwise)
    exp = O.roi_align_rotated_forward(x, rois, 7, 7, 0.25, ratio, aligned, clockwise)
    got = y.detach().cpu().numpy()
    assert np.allclose(got, exp, rtol=1e-5, atol=1e-6), np.abs(got - exp).max()
    go = rng.randn(*exp.shape).astype(np.float32)
    y.backward(dev(go))
    gexp = O.roi_align_rotated_backward(go, rois, x.shape, 7, 7, 0.25, ratio, aligned, clockwise)
    ggot = xt.grad.cpu().numpy()
    assert np.allclose(ggot, gexp, rtol=1e-4, atol=1e-4), np.abs(ggot - gexp).max()


def test_roi_align_rotated_config_shape():
    """BASELINE shape: 512 rois x C256 x 7x7 x s2 on the stride-4 level of one 1024^2 image (256x256 map)."""
    ops, O = _ops(), _oracle()
    rng = np.random.RandomState(3)
    x = rng.randn(1, 256, 64, 64).astype(np.float32)  # oracle-sized map; the full 256x256 map is in bench
    rois = synth.rois_for_level(512, 5, batch=1, extent=256.0, wh=(8.0, 200.0))
    layer = ops.RoIAlignRotated(out_size=7, spatial_scale=0.25, sample_num=2, clockwise=True)
    y = layer(dev(x), dev(rois)).cpu().numpy()
    exp = O.roi_align_rotated_forward(x, rois, 7, 7, 0.25, 2, True, True)
    assert np.allclose(y, exp, rtol=1e-5, atol=1e-6)


def test_cpu_tensors_are_rejected():
    from sm3det_amd import mmcv_ext
    with pytest.raises(RuntimeError):
        mmcv_ext.box_iou_rotated(torch.zeros(1, 5), torch.zeros(1, 5), torch.zeros(1), 0, False)
    with pytest.raises(NotImplementedError):
        mmcv_ext.roi_align_forward()


# ----------------------------------------------------------------------------------- the shapes bench.py times
def test_nms_rotated_bench_shape_10000_vs_oracle_and_compiled_reference():
    """the exact inputs of bench.py's `nms_rotated_10000` line: keep list bit-exact vs the C oracle and vs the
    reference's own CPU op (oracle/_ref) when its .so travelled."""
    ops, O = _ops(), _oracle()
    d, s = synth.rotated_boxes(10000, 7), synth.unique_scores(10000, 8)
    _, keep = ops.nms_rotated(dev(d), dev(s), 0.1)
    exp = O.nms_rotated(d, s, 0.1)
    assert np.array_equal(keep.cpu().numpy(), exp), (len(exp), keep.numel())
    ref = _ref_or_none()
    if ref is not None:
        kr = ref.nms_rotated_cpu(torch.from_numpy(d), torch.from_numpy(s), 0.1)
        assert np.array_equal(keep.cpu().numpy(), kr.numpy())
    # clustered (dense-overlap) variant of the same size
    d, s = synth.rotated_boxes(10000, 9, cluster=True), synth.unique_scores(10000, 10)
    _, keep = ops.nms_rotated(dev(d), dev(s), 0.1)
    assert np.array_equal(keep.cpu().numpy(), O.nms_rotated(d, s, 0.1))


@pytest.mark.parametrize('channels_last', [False, True])
def test_roi_align_rotated_bench_shape_256x256_level(channels_last):
    """bench.py's RoIAlignRotated lines: 512 RoIs x 256 ch x 7x7 x s2 on the REAL stride-4 level (256x256 map) of one
    1024^2 image, forward and backward, NCHW and NHWC features."""
    ops, O = _ops(), _oracle()
    rng = np.random.RandomState(3)
    x = rng.randn(1, 256, 256, 256).astype(np.float32)
    rois = synth.rois_for_level(512, 6, batch=1, extent=1024.0)
    xt = dev(x).requires_grad_(True)
    xin = xt.contiguous(memory_format=torch.channels_last) if channels_last else xt
    layer = ops.RoIAlignRotated(output_size=7, spatial_scale=0.25, sampling_ratio=2, clockwise=True)
    y = layer(xin, dev(rois))
    exp = O.roi_align_rotated_forward(x, rois, 7, 7, 0.25, 2, True, True)
    got = y.detach().cpu().numpy()
    assert np.allclose(got, exp, rtol=1e-5, atol=1e-6), np.abs(got - exp).max()
    go = rng.randn(*exp.shape).astype(np.float32)
    y.backward(dev(go))
    gexp = O.roi_align_rotated_backward(go, rois, x.shape, 7, 7, 0.25, 2, True, True)
    ggot = xt.grad.cpu().numpy()
    assert np.allclose(ggot, gexp, rtol=1e-4, atol=1e-4), np.abs(ggot - gexp).max()


@pytest.mark.parametrize('hw,c', [(64, 64), (16, 8)])  # NHWC-scratch path (H*W >= 1024, C >= 32) and the direct NCHW path
def test_roi_align_rotated_backward_accumulates_into_grad_input(hw, c):
    """mmcv._ext.roi_align_rotated_backward ADDS into grad_input (atomicAdd in the reference kernel,
    roi_align_rotated_cuda_kernel.cuh:129-200): a caller-provided non-zero tensor must survive on both internal paths."""
    from sm3det_amd import mmcv_ext
    O = _oracle()
    rng = np.random.RandomState(5)
    rois = synth.rois_for_level(40, 9, batch=1, extent=float(hw * 4))
    go = rng.randn(40, c, 7, 7).astype(np.float32)
    base = rng.randn(1, c, hw, hw).astype(np.float32)
    gi = dev(base.copy())
    mmcv_ext.roi_align_rotated_backward(dev(go), dev(rois), gi, 7, 7, 0.25, 2, True, True)
    exp = base + O.roi_align_rotated_backward(go, rois, base.shape, 7, 7, 0.25, 2, True, True)
    assert np.allclose(gi.cpu().numpy(), exp, rtol=1e-4, atol=1e-4)


def test_box_iou_rotated_bench_shape_2000x64():
    ops, O = _ops(), _oracle()
    b1, b2 = synth.rotated_boxes(2000, 0), synth.rotated_boxes(64, 1)
    got = ops.box_iou_rotated(dev(b1), dev(b2)).cpu().numpy()
    exp = O.box_iou_rotated(b1, b2, 0)
    assert np.abs(got - exp).max() <= 1e-6 and np.array_equal(got > 0, exp > 0)


def test_nms_rotated_threshold_equality_follows_cpu_path():
    """exact duplicates at iou_threshold = 1.0: the reference CPU op suppresses with `>=` (cpu/nms_rotated.cpp), its CUDA
    op with `>`; this implementation follows the CPU path (documented in sm3det_amd/mmcv_ext.py)."""
    ops, O = _ops(), _oracle()
    d = np.array([[50, 50, 20, 10, 0.3], [50, 50, 20, 10, 0.3], [200, 200, 30, 30, 0.0], [200, 200, 30, 30, 0.0],
                  [400, 100, 8, 40, -0.7]], np.float32)
    s = np.array([0.9, 0.8, 0.7, 0.6, 0.5], np.float32)
    _, keep = ops.nms_rotated(dev(d), dev(s), 1.0)
    exp = O.nms_rotated(d, s, 1.0)
    assert np.array_equal(keep.cpu().numpy(), exp)
    # the axis-aligned duplicate has IoU exactly 1.0 and is suppressed by `>=` (a `>` rule would keep it); the rotated
    # duplicate's IoU rounds just below 1.0 in fp32 and survives under either rule
    assert keep.cpu().tolist() == [0, 1, 2, 4]
    ref = _ref_or_none()
    if ref is not None:
        assert ref.nms_rotated_cpu(torch.from_numpy(d), torch.from_numpy(s), 1.0).tolist() == [0, 1, 2, 4]


# ----------------------------------------------------------------------------------- round 4: tiled backward, vector forward
@pytest.mark.parametrize('C,hw', [(256, 64), (64, 40), (96, 33), (320, 24)])
def test_roi_align_rotated_tiled_backward_equals_the_atomic_form(C, hw, monkeypatch):
    """SM3_ROI_BWD=tiled (counting sort by pixel + gather per 8x8 tile, the default on NHWC maps) against SM3_ROI_BWD=atomic
    (the scatter kernel) and the oracle, batch 2, map sizes that are not multiples of the tile, channel counts below / at /
    above the 256-channel pass of a wave; accumulation into a non-zero grad_input."""
    from sm3det_amd import mmcv_ext
    O = _oracle()
    B = 2
    rois = synth.rois_for_level(300, 21, batch=B, extent=hw * 4.0, wh=(4.0, hw * 2.0))
    go = np.random.RandomState(5).randn(300, C, 7, 7).astype(np.float32)
    base = np.random.RandomState(6).randn(B, C, hw, hw).astype(np.float32)
    exp = base + O.roi_align_rotated_backward(go, rois, base.shape, 7, 7, 0.25, 2, True, True)
    outs = {}
    for mode in ('tiled', 'atomic'):
        monkeypatch.setenv('SM3_ROI_BWD', mode)
        gi = dev(base).contiguous(memory_format=torch.channels_last)
        mmcv_ext.roi_align_rotated_backward(dev(go), dev(rois), gi, 7, 7, 0.25, 2, True, True)
        outs[mode] = gi.cpu().numpy()
        assert np.allclose(outs[mode], exp, rtol=1e-4, atol=1e-4), (mode, np.abs(outs[mode] - exp).max())
    assert np.allclose(outs['tiled'], outs['atomic'], rtol=1e-4, atol=1e-4)


def test_roi_align_rotated_tiled_backward_overwrite_mode_writes_every_pixel():
    """overwrite != 0 (what the pyramid extractor and the NCHW wrapper use: no zero-fill pass): two levels whose sizes are
    not multiples of the tile, maps poisoned with NaN beforehand -> exactly the oracle's gradient of zero-filled maps,
    zeros included; the RoI set leaves whole tiles (and one whole level of batch 1) untouched."""
    import ctypes
    from sm3det_amd import _lib
    O = _oracle()
    B, C, n = 2, 64, 120
    shapes = [(B, C, 52, 44), (B, C, 26, 22)]
    strides = [4.0, 8.0]
    rois = synth.rois_for_level(n, 23, batch=1, extent=120.0, wh=(6.0, 200.0))  # batch index 0 only, upper-left corner
    go = np.random.RandomState(7).randn(n, C, 7, 7).astype(np.float32)
    # the oracle, level by level (map_roi_levels, finest_scale 56: single_level_roi_extractor.py:66-84)
    scale = np.sqrt(rois[:, 3] * rois[:, 4])
    lvl = np.clip(np.floor(np.log2(scale / 56.0 + 1e-6)), 0, 1).astype(int)
    assert set(lvl.tolist()) == {0, 1}
    exp = [O.roi_align_rotated_backward(go[lvl == l], rois[lvl == l], shapes[l], 7, 7, 1.0 / strides[l], 2, True, True)
           for l in range(2)]
    maps = [torch.full((B, s[2], s[3], C), float('nan'), device='cuda') for s in shapes]  # NHWC memory
    L = lib = _lib.lib()
    hs, ws = (ctypes.c_int * 2)(52, 26), (ctypes.c_int * 2)(44, 22)
    sc = (ctypes.c_float * 2)(0.25, 0.125)
    ptrs = (ctypes.c_void_p * 2)(*[m.data_ptr() for m in maps])
    nb = lib.sm3_roi_align_rotated_backward_tiled_workspace_bytes(n, B, C, 7, 7, 2, hs, ws, 2)
    wsp = _lib.workspace(nb, maps[0].device)
    g, r = dev(go), dev(rois)
    _lib.check(L.sm3_roi_align_rotated_backward_tiled(g.data_ptr(), r.data_ptr(), ptrs, hs, ws, sc, 2, 56.0, n, B, C, 7, 7, 2,
                                                      1, 1, 1, wsp.data_ptr(), nb, _lib.stream_ptr()), 'tiled')
    for l in range(2):
        got = maps[l].permute(0, 3, 1, 2).cpu().numpy()
        assert np.isfinite(got).all()
        assert np.allclose(got, exp[l], rtol=1e-4, atol=1e-4), np.abs(got - exp[l]).max()
        assert (got[1] == 0).all()  # no RoI on image 1
    # n_rois == 0 in overwrite mode: the maps are zeroed
    maps[0].fill_(float('nan'))
    _lib.check(L.sm3_roi_align_rotated_backward_tiled(g.data_ptr(), r.data_ptr(), ptrs, hs, ws, sc, 2, 56.0, 0, B, C, 7, 7, 2,
                                                      1, 1, 1, wsp.data_ptr(), nb, _lib.stream_ptr()), 'tiled')
    assert (maps[0] == 0).all()


def test_roi_align_rotated_vector_forward_is_bit_identical_to_the_scalar_kernel():
    """NHWC channel-vector forward (C % 4 == 0, 16-byte gathers) vs the NCHW scalar kernel on the same taps: same fp32
    operation order per channel -> identical bits (small RoI count keeps the NCHW call on the scalar path)."""
    ops = _ops()
    x = torch.randn(2, 256, 48, 48, device='cuda')
    rois = dev(synth.rois_for_level(5, 22, batch=2, extent=192.0, wh=(4.0, 120.0)))  # 5 RoIs: the NCHW call stays scalar
    y_nchw = ops.roi_align_rotated(x, rois, 7, 0.25, 2, True, True)
    y_nhwc = ops.roi_align_rotated(x.contiguous(memory_format=torch.channels_last), rois, 7, 0.25, 2, True, True)
    assert torch.equal(y_nchw, y_nhwc)


_TWO_PROC_WORKER = r'''
import sys, time
import numpy as np, torch
sys.path.insert(0, %(root)r)
from sm3det_amd import mmcv_ops, assign
from oracle import ops_oracle as O
from tests import synth
seed = int(sys.argv[1])
b1, b2 = synth.rotated_boxes(600, seed, cluster=True), synth.rotated_boxes(96, seed + 1, cluster=True)
want = O.box_iou_rotated(b1, b2)
s = synth.unique_scores(600, seed + 2)
keep_want = O.nms_rotated(b1, s, 0.1)
d1, d2, ds = torch.from_numpy(b1).cuda(), torch.from_numpy(b2).cuda(), torch.from_numpy(s).cuda()
t0, n = time.time(), 0
while time.time() - t0 < 4.0:  # both processes hammer the rotated-IoU kernels for the same few seconds
    got = mmcv_ops.box_iou_rotated(d1, d2)
    _, keep = mmcv_ops.nms_rotated(d1, ds, 0.1)
    res = assign.MaxIoUAssigner(0.5, 0.4, 0.3, iou_calculator=dict(type='RBboxOverlaps2D')).assign(d1, d2[:8])
    torch.cuda.synchronize()
    assert np.abs(got.cpu().numpy() - want).max() <= 1e-6
    assert np.array_equal(keep.cpu().numpy(), keep_want)
    assert res.gt_inds.numel() == 600
    n += 1
print('ok', n)
'''


def test_two_processes_share_the_gpu_while_running_the_rotated_iou_kernels(tmp_path):
    """Round 5 saw GPU faults when two processes ran the rotated-IoU kernels (box_iou_rotated, nms_rotated's mask kernel,
    the rotated MaxIoU assignment) on ONE GPU at the same time; the one thing those kernels shared was 528 bytes of private
    (scratch) memory per lane.  Their polygon scratch now lives in LDS (no private segment at all, checked by
    tests/test_abi.py on the code object); this test runs two such processes concurrently and checks both against the oracle."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'worker.py'
    script.write_text(_TWO_PROC_WORKER % dict(root=root))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    procs = [subprocess.Popen([sys.executable, str(script), str(10 * i)], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                              env=env, cwd=root) for i in range(2)]
    outs = [p.communicate(timeout=300) for p in procs]
    for p, (o, e) in zip(procs, outs):
        assert p.returncode == 0, (p.returncode, o.decode()[-500:], e.decode()[-2000:])
        assert o.decode().strip().startswith('ok')
