"""CPU tier: pins oracle/ops_oracle.c (plain-C restatement) against
 (a) the mmcv golden vectors (tests/golden_vectors.py, values from mmcv/tests/test_ops/*),
 (b) oracle/_ref = the reference's own C++ compiled by oracle/build_ref.py -- bit-exact."""
import numpy as np
import pytest
import torch

from oracle import build_ref
from oracle import ops_oracle as O
from tests import golden_vectors as GV
from tests import synth


def _ref():
    try:
        build_ref.build()
        return build_ref.load_ref()
    except Exception as e:  # pragma: no cover
        pytest.skip(f'oracle/_ref unavailable: {e}')


# ---------------- golden vectors ----------------
def test_box_iou_rotated_golden():
    g = GV.BOX_IOU_ROTATED
    assert np.allclose(O.box_iou_rotated(g['boxes1'], g['boxes2']), g['ious'], atol=1e-4)
    assert np.allclose(O.box_iou_rotated(g['boxes1'], g['boxes2'], aligned=True), np.diag(g['ious']),
                       atol=1e-4)


def test_nms_rotated_golden():
    g = GV.NMS_ROTATED
    assert O.nms_rotated(g['dets'][:, :5], g['dets'][:, 5], g['thr']).tolist() == g['keep']
    # 6-column dets (label column appended) -> CPU path ignores the label (cpu/nms_rotated.cpp:47-48)
    d6 = np.concatenate([g['dets'][:, :5], g['labels'][:, None]], 1)
    assert O.nms_rotated(d6, g['dets'][:, 5], g['thr']).tolist() == g['keep']


def test_nms_golden():
    g = GV.NMS
    assert O.nms(g['boxes'], g['scores'], g['thr'], 0).tolist() == g['keep']


@pytest.mark.parametrize('case', range(len(GV.ROI_ALIGN_ROTATED_CASES)))
def test_roi_align_rotated_golden(case):
    x, rois, out, grad = GV.ROI_ALIGN_ROTATED_CASES[case]
    x = np.array(x, np.float32)
    rois = np.array(rois, np.float32)
    y = O.roi_align_rotated_forward(x, rois, 2, 2, 1.0, 2, True, False)
    assert np.allclose(y, np.array(out, np.float32), atol=1e-3)
    g = O.roi_align_rotated_backward(np.ones_like(y), rois, x.shape, 2, 2, 1.0, 2, True, False)
    assert np.allclose(g, np.array(grad, np.float32), atol=1e-3)


# ---------------- bit-exact vs the compiled reference ----------------
@pytest.mark.parametrize('seed,cluster', [(0, False), (1, True), (2, True)])
def test_box_iou_rotated_vs_ref(seed, cluster):
    ref = _ref()
    b1 = synth.rotated_boxes(300, seed, extent=256.0, cluster=cluster)
    b2 = synth.rotated_boxes(64, seed + 100, extent=256.0, cluster=cluster)
    for mode in (0, 1):
        out = torch.zeros(300 * 64)
        ref.box_iou_rotated(torch.from_numpy(b1), torch.from_numpy(b2), out, mode, False)
        mine = O.box_iou_rotated(b1, b2, mode, False)
        assert np.array_equal(mine.reshape(-1), out.numpy()), np.abs(mine.reshape(-1) - out.numpy()).max()
        assert (mine > 0).any()


def test_box_iou_rotated_degenerate_vs_ref():
    ref = _ref()
    b1, b2 = synth.degenerate_rotated_pairs()
    out = torch.zeros(len(b1))
    ref.box_iou_rotated(torch.from_numpy(b1), torch.from_numpy(b2), out, 0, True)
    mine = O.box_iou_rotated(b1, b2, 0, True)
    assert np.array_equal(mine, out.numpy()), (mine, out.numpy())
    full = torch.zeros(len(b1) * len(b2))
    ref.box_iou_rotated(torch.from_numpy(b1), torch.from_numpy(b2), full, 0, False)
    assert np.array_equal(O.box_iou_rotated(b1, b2).reshape(-1), full.numpy())


@pytest.mark.parametrize('n,thr,cluster', [(500, 0.1, False), (700, 0.1, True), (400, 0.5, True), (1, 0.3, False)])
def test_nms_rotated_vs_ref(n, thr, cluster):
    ref = _ref()
    d = synth.rotated_boxes(n, 3, extent=512.0, cluster=cluster)
    s = synth.unique_scores(n, 4)
    keep_ref = ref.nms_rotated_cpu(torch.from_numpy(d), torch.from_numpy(s), thr).numpy()
    keep = O.nms_rotated(d, s, thr)
    assert np.array_equal(keep, keep_ref)
    assert 0 < len(keep) <= n


@pytest.mark.parametrize('n,thr,offset,cluster', [(2000, 0.8, 0, True), (1500, 0.6, 1, True), (300, 0.3, 0, False)])
def test_nms_vs_ref(n, thr, offset, cluster):
    ref = _ref()
    b = synth.hboxes(n, 5, cluster=cluster)
    s = synth.unique_scores(n, 6)
    keep_ref = ref.nms(torch.from_numpy(b), torch.from_numpy(s), thr, offset).numpy()
    keep = O.nms(b, s, thr, offset)
    assert np.array_equal(keep, keep_ref)


def test_empty_inputs():
    assert O.nms(np.zeros((0, 4), np.float32), np.zeros((0,), np.float32), 0.5).size == 0
    assert O.nms_rotated(np.zeros((0, 5), np.float32), np.zeros((0,), np.float32), 0.5).size == 0
    assert O.box_iou_rotated(np.zeros((0, 5), np.float32), np.zeros((3, 5), np.float32)).shape == (0, 3)


@pytest.mark.parametrize('aligned,clockwise,ratio', [(True, True, 2), (True, False, 0), (False, True, 2)])
def test_roi_align_rotated_vs_ref(aligned, clockwise, ratio):
    ref = _ref()
    rng = np.random.RandomState(0)
    x = rng.randn(2, 5, 40, 48).astype(np.float32)
    rois = synth.rois_for_level(60, 1, batch=2, extent=48 * 4.0, wh=(4.0, 120.0))
    # a few rois hanging outside the map and tiny ones
    rois[0, 1:3] = [-20, -20]
    rois[1, 1:3] = [400, 300]
    rois[2, 3:5] = [0.5, 0.5]
    xt, rt = torch.from_numpy(x), torch.from_numpy(rois)
    out = torch.zeros(60, 5, 7, 7)
    ref.roi_align_rotated_forward(xt, rt, out, 7, 7, 0.25, ratio, aligned, clockwise)
    y = O.roi_align_rotated_forward(x, rois, 7, 7, 0.25, ratio, aligned, clockwise)
    assert np.array_equal(y, out.numpy()), np.abs(y - out.numpy()).max()
    go = rng.randn(60, 5, 7, 7).astype(np.float32)
    gin = torch.zeros(2, 5, 40, 48)
    ref.roi_align_rotated_backward(torch.from_numpy(go), rt, gin, 7, 7, 0.25, ratio, aligned, clockwise)
    g = O.roi_align_rotated_backward(go, rois, x.shape, 7, 7, 0.25, ratio, aligned, clockwise)
    # the reference accumulates in (n,c,ph,pw) index order, and so does the oracle: bit-exact
    assert np.array_equal(g, gin.numpy()), np.abs(g - gin.numpy()).max()
