"""The drop-in boundary proven from the REFERENCE's side: the reference's own, unmodified python
wrappers ``mmcv/mmcv/ops/{box_iou_rotated,nms,roi_align_rotated,deform_conv}.py`` (loaded by oracle/ref_mmcv_ops.py from
/root/reference, or from the bytecode oracle/build_ref.py compiled from them where that tree is absent) bind to
``sm3det_amd.mmcv_ext`` installed as ``mmcv._ext`` through their own ``ext_loader.load_ext``.

* CPU tier: every native call a reference wrapper issues (name, positional and keyword arguments, tensor values) is
  recorded and must equal, call for call, what this package's mirror wrappers (sm3det_amd/mmcv_ops.py,
  mmcv_deform_conv.py -- the ones all other GPU tests go through) issue for the same inputs.
* GPU tier: the reference wrappers themselves run on the MI355X kernels and are checked against the C oracle
  (bit-exact keep lists, IoU <= 1e-6, RoIAlign 1e-5 / 1e-4, DeformConv2d vs the compiled reference)."""
import types

import numpy as np
import pytest
import torch

from oracle import ref_mmcv_ops as R
from tests import synth

needs_ref = pytest.mark.skipif(not R.available(), reason='reference wrappers neither in /root/reference nor oracle/_ref/pyc')


class _Recorder(types.ModuleType):
    """stands in for mmcv._ext: records calls, fills / returns plausible outputs"""

    def __init__(self):
        super().__init__('mmcv._ext')
        self.calls = []
        from sm3det_amd import mmcv_ext
        for name in dir(mmcv_ext):
            if not name.startswith('_') and callable(getattr(mmcv_ext, name)) and name not in ('install_as_mmcv_ext',):
                setattr(self, name, self._make(name))

    def _make(self, name):
        def fn(*a, **k):
            self.calls.append((name, tuple(self._freeze(x) for x in a), {kk: self._freeze(v) for kk, v in k.items()}))
            if name in ('nms', 'nms_rotated'):
                return torch.arange(min(3, a[0].size(0)), dtype=torch.long)
            return None
        return fn

    @staticmethod
    def _freeze(x):
        return ('T', tuple(x.shape), str(x.dtype), x.detach().clone()) if isinstance(x, torch.Tensor) else x


def _same(a, b):
    if isinstance(a, tuple) and a and a[0] == 'T':
        return isinstance(b, tuple) and b[0] == 'T' and a[1] == b[1] and a[2] == b[2] and torch.equal(a[3], b[3])
    if isinstance(a, float) or isinstance(b, float):
        return float(a) == float(b)
    return a == b


_UNINITIALISED_OUTPUTS = {('deform_conv_forward', 3)}  # deform_conv.py:79 `output = input.new_empty(...)`


def _assert_same_calls(ref_calls, our_calls):
    assert [c[0] for c in ref_calls] == [c[0] for c in our_calls]
    for (n, ra, rk), (_, oa, ok) in zip(ref_calls, our_calls):
        assert len(ra) == len(oa) and set(rk) == set(ok), (n, len(ra), len(oa), sorted(rk), sorted(ok))
        for i, (x, y) in enumerate(zip(ra, oa)):
            if (n, i) in _UNINITIALISED_OUTPUTS:  # `input.new_empty(...)` buffers: same shape / dtype, contents undefined
                assert x[1:3] == y[1:3], (n, 'positional', i)
                continue
            assert _same(x, y), (n, 'positional', i)
        for kk in rk:
            assert _same(rk[kk], ok[kk]), (n, kk, rk[kk], ok[kk])


@needs_ref
def test_reference_wrappers_and_mirror_wrappers_issue_identical_native_calls(monkeypatch):
    from sm3det_amd import mmcv_deform_conv, mmcv_ops
    rec_ref, rec_our = _Recorder(), _Recorder()
    ref = R.load(rec_ref)
    monkeypatch.setattr(mmcv_ops, 'ext_module', rec_our)
    monkeypatch.setattr(mmcv_deform_conv, 'ext_module', rec_our)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))  # noqa: E731
    b1, b2 = T(synth.rotated_boxes(7, 0)), T(synth.rotated_boxes(5, 1))
    sc = T(synth.unique_scores(7, 2))

    def both(ref_fn, our_fn, *a, **k):
        n0, n1 = len(rec_ref.calls), len(rec_our.calls)
        r, o = ref_fn(*a, **k), our_fn(*a, **k)
        _assert_same_calls(rec_ref.calls[n0:], rec_our.calls[n1:])
        assert len(rec_ref.calls) > n0
        return r, o
    # box_iou_rotated: modes, aligned, ccw
    for kw in (dict(), dict(mode='iof'), dict(aligned=True), dict(clockwise=False)):
        bb2 = b2 if not kw.get('aligned') else b1
        both(ref['box_iou_rotated'].box_iou_rotated, mmcv_ops.box_iou_rotated, b1.clone(), bb2.clone(), **kw)
    # nms: plain, offset, deprecated iou_thr kwarg, score threshold + max_num
    hb, hs = T(synth.hboxes(9, 3)), T(synth.unique_scores(9, 4))
    both(ref['nms'].nms, mmcv_ops.nms, hb, hs, iou_threshold=0.5)
    both(ref['nms'].nms, mmcv_ops.nms, hb, hs, 0.3, 1)
    both(ref['nms'].nms, mmcv_ops.nms, hb, hs, iou_thr=0.6)
    both(ref['nms'].nms, mmcv_ops.nms, hb, hs, iou_threshold=0.5, score_threshold=0.4, max_num=2)
    # nms_rotated: with / without labels, ccw
    both(ref['nms'].nms_rotated, mmcv_ops.nms_rotated, b1.clone(), sc, 0.1)
    both(ref['nms'].nms_rotated, mmcv_ops.nms_rotated, b1.clone(), sc, 0.1, torch.arange(7) % 3)
    both(ref['nms'].nms_rotated, mmcv_ops.nms_rotated, b1.clone(), sc, 0.1, clockwise=False)
    # batched_nms over both NMS flavours
    ids = torch.arange(9) % 2
    both(ref['nms'].batched_nms, mmcv_ops.batched_nms, hb, hs, ids, dict(type='nms', iou_threshold=0.5))
    both(ref['nms'].batched_nms, mmcv_ops.batched_nms, b1, sc, torch.arange(7) % 2,
         dict(type='nms_rotated', iou_threshold=0.1))
    # RoIAlignRotated module (deprecated ctor aliases as in main_SM3Det.py) forward + backward
    x = torch.randn(2, 4, 8, 8)
    rois = T(synth.rois_for_level(6, 5, batch=2, extent=32.0, wh=(2.0, 20.0)))
    for mod_ref, mod_our in ((ref['roi_align_rotated'].RoIAlignRotated, mmcv_ops.RoIAlignRotated),):
        lr = mod_ref(out_size=7, spatial_scale=0.25, sample_num=2, clockwise=True)
        lo = mod_our(out_size=7, spatial_scale=0.25, sample_num=2, clockwise=True)
        n0, n1 = len(rec_ref.calls), len(rec_our.calls)
        xr, xo = x.clone().requires_grad_(True), x.clone().requires_grad_(True)
        yr, yo = lr(xr, rois), lo(xo, rois)
        go = torch.randn_like(yr)
        yr.backward(go)
        yo.backward(go)
        _assert_same_calls(rec_ref.calls[n0:], rec_our.calls[n1:])
        assert [c[0] for c in rec_ref.calls[n0:]] == ['roi_align_rotated_forward', 'roi_align_rotated_backward']
    # DeformConv2d functional: forward + both backward entry points
    xd = torch.randn(2, 4, 6, 6)
    off = torch.randn(2, 18, 6, 6)
    w = torch.randn(8, 4, 3, 3)
    n0, n1 = len(rec_ref.calls), len(rec_our.calls)
    for fn in (ref['deform_conv'].deform_conv2d, mmcv_deform_conv.deform_conv2d):
        a, o_, w_ = xd.clone().requires_grad_(True), off.clone().requires_grad_(True), w.clone().requires_grad_(True)
        y = fn(a, o_, w_, 1, 1, 1, 1, 1, False, 2)
        y.backward(torch.ones_like(y))
    _assert_same_calls(rec_ref.calls[n0:], rec_our.calls[n1:])
    assert [c[0] for c in rec_ref.calls[n0:]] == ['deform_conv_forward', 'deform_conv_backward_input',
                                                  'deform_conv_backward_parameters']


@needs_ref
@pytest.mark.gpu
def test_reference_wrappers_run_on_the_mi355x_kernels_and_match_the_oracle():
    from oracle import ops_oracle as O
    ref = R.load()  # mmcv._ext := sm3det_amd.mmcv_ext
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a)).cuda()  # noqa: E731
    # box_iou_rotated (reference wrapper) vs C oracle
    b1, b2 = synth.rotated_boxes(500, 0, cluster=True), synth.rotated_boxes(64, 1, cluster=True)
    got = ref['box_iou_rotated'].box_iou_rotated(dev(b1), dev(b2)).cpu().numpy()
    exp = O.box_iou_rotated(b1, b2, 0)
    assert np.abs(got - exp).max() <= 1e-6 and np.array_equal(got > 0, exp > 0)
    # nms / nms_rotated / batched_nms (reference wrappers): keep lists bit-exact
    hb, hs = synth.hboxes(3000, 4, cluster=True), synth.unique_scores(3000, 5)
    dets, keep = ref['nms'].nms(dev(hb), dev(hs), iou_threshold=0.6)
    assert np.array_equal(keep.cpu().numpy(), O.nms(hb, hs, 0.6, 0)) and dets.shape[1] == 5
    d, s = synth.rotated_boxes(2000, 2, cluster=True), synth.unique_scores(2000, 3)
    dets, keep = ref['nms'].nms_rotated(dev(d), dev(s), 0.1)
    assert np.array_equal(keep.cpu().numpy(), O.nms_rotated(d, s, 0.1)) and dets.shape[1] == 6
    # RoIAlignRotated (reference nn.Module with the config's deprecated kwargs) forward + backward
    rng = np.random.RandomState(0)
    x = rng.randn(2, 16, 40, 48).astype(np.float32)
    rois = synth.rois_for_level(80, 1, batch=2, extent=48 * 4.0, wh=(4.0, 120.0))
    layer = ref['roi_align_rotated'].RoIAlignRotated(out_size=7, spatial_scale=0.25, sample_num=2, clockwise=True)
    xt = dev(x).requires_grad_(True)
    y = layer(xt, dev(rois))
    exp = O.roi_align_rotated_forward(x, rois, 7, 7, 0.25, 2, True, True)
    assert np.allclose(y.detach().cpu().numpy(), exp, rtol=1e-5, atol=1e-6)
    go = rng.randn(*exp.shape).astype(np.float32)
    y.backward(dev(go))
    gexp = O.roi_align_rotated_backward(go, rois, x.shape, 7, 7, 0.25, 2, True, True)
    assert np.allclose(xt.grad.cpu().numpy(), gexp, rtol=1e-4, atol=1e-4)
    # DeformConv2d (reference functional) vs the compiled reference CPU op, when its .so travelled
    try:
        from oracle import build_ref
        cref = build_ref.load_ref()
    except Exception:
        return
    g = torch.Generator().manual_seed(0)
    xd, off = torch.randn(2, 32, 20, 24, generator=g), torch.randn(2, 18, 20, 24, generator=g) * 2
    w = torch.randn(32, 32, 3, 3, generator=g) * 0.2
    out_r = torch.zeros(2, 32, 20, 24)
    cref.deform_conv_forward(xd, w, off, out_r, torch.zeros(0), torch.zeros(0), 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 2)
    out = ref['deform_conv'].deform_conv2d(xd.cuda(), off.cuda(), w.cuda(), 1, 1, 1, 1, 1, False, 2)
    assert float((out.cpu() - out_r).abs().max() / out_r.abs().max()) < 1e-4
