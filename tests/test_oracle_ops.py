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


@pytest.mark.parametrize('seed', [7, 8, 9])
def test_circumscribed_circle_pretest_is_conservative(seed):
    """The GPU kernels skip the polygon clipping for pairs whose circumscribed circles are apart AND whose boxes are not
    thin (ops_rotated.hip `may_intersect`) and write IoU = +0.0 for them.  Property pinned here against the oracle
    (bit-exact with the compiled reference): every pair the pre-test rejects -- evaluated in fp32 exactly as the kernel
    does -- has IoU == 0.0 in the reference arithmetic, on random, near-touching, far-away and thin / degenerate boxes.
    (Thin boxes matter: below ~1e-7 of the coordinates an edge vector collapses to zero in fp32 and the reference reports
    IoU = 1 / inf / negative for boxes that are far apart; the shortcut must leave those to the full computation.)"""
    f = np.float32
    rng = np.random.default_rng(seed)
    n = 150000
    b1 = np.stack([rng.uniform(0, 512, n), rng.uniform(0, 512, n), rng.uniform(0.5, 160, n), rng.uniform(0.5, 160, n),
                   rng.uniform(-np.pi, np.pi, n)], 1).astype(f)
    b2 = b1.copy()
    b2[:, 2:4] = rng.uniform(0.5, 160, (n, 2)).astype(f)
    b2[:, 4] = rng.uniform(-np.pi, np.pi, n).astype(f)
    # a third of the pairs get one thin side, log-uniform from 1e-10 to 1 (the collapse happens around 1e-5 here)
    thin = rng.random(n) < 0.33
    side = rng.integers(0, 2, n)
    which = rng.integers(0, 2, n)
    t = (10.0 ** rng.uniform(-10, 0, n)).astype(f)
    for bx, sel in ((b1, thin & (which == 0)), (b2, thin & (which == 1))):
        bx[sel & (side == 0), 2] = t[sel & (side == 0)]
        bx[sel & (side == 1), 3] = t[sel & (side == 1)]
    b2[:300, 2] = 0
    b1[300:600, 3] = 0
    # second box at a distance around the sum of the circumscribed radii (the interesting band), any direction
    r1 = 0.5 * np.sqrt(b1[:, 2].astype(np.float64) ** 2 + b1[:, 3].astype(np.float64) ** 2)
    r2 = 0.5 * np.sqrt(b2[:, 2].astype(np.float64) ** 2 + b2[:, 3].astype(np.float64) ** 2)
    dist = (r1 + r2) * rng.choice([0.2, 0.9, 0.99, 1.0, 1.004, 1.006, 1.02, 1.5, 4.0, 20.0], n)
    ang = rng.uniform(0, 2 * np.pi, n)
    b2[:, 0] = (b1[:, 0] + dist * np.cos(ang)).astype(f)
    b2[:, 1] = (b1[:, 1] + dist * np.sin(ang)).astype(f)
    far = slice(n - 4000, n)  # far-away coordinates: cancellation in the centre shift
    b1[far, :2] += f(3e5)
    b2[far, :2] += f(3e5)

    def pretest(a, b):  # ops_rotated.hip: circum_radius / min_extent / may_intersect, fp32, same operation order
        ra = f(0.5) * np.sqrt(a[:, 2] * a[:, 2] + a[:, 3] * a[:, 3], dtype=f)
        rb = f(0.5) * np.sqrt(b[:, 2] * b[:, 2] + b[:, 3] * b[:, 3], dtype=f)
        ea, eb = np.minimum(np.abs(a[:, 2]), np.abs(a[:, 3])), np.minimum(np.abs(b[:, 2]), np.abs(b[:, 3]))
        dx, dy, rs = a[:, 0] - b[:, 0], a[:, 1] - b[:, 1], ra + rb
        apart = dx * dx + dy * dy > rs * rs * f(1.01) + f(1e-12)
        solid = np.minimum(ea, eb) >= f(1e-3) * (np.abs(dx) + np.abs(dy) + rs)
        return ~(apart & solid)

    keep = pretest(b1, b2)
    iou = O.box_iou_rotated(b1, b2, 0, True)
    assert 0.2 < keep.mean() < 0.9  # both outcomes well represented
    assert (iou[~keep] == 0.0).all(), (np.abs(iou[~keep]).max(), int((iou[~keep] != 0).sum()))
    assert (iou[keep] > 0).any()
    # the reference does report intersections for far-apart thin boxes: the reason for the `solid` clause
    ra = 0.5 * np.sqrt(b1[:, 2] ** 2 + b1[:, 3] ** 2) + 0.5 * np.sqrt(b2[:, 2] ** 2 + b2[:, 3] ** 2)
    apart = (b1[:, 0] - b2[:, 0]) ** 2 + (b1[:, 1] - b2[:, 1]) ** 2 > 1.02 * ra ** 2
    assert (iou[apart] != 0).any()
    iof = O.box_iou_rotated(b1, b2, 1, True)
    assert (iof[~keep] == 0.0).all()
