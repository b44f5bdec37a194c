"""CPU tier: `sm3det_amd.detector.TriSourceDetector` -- (i) construction from the unchanged `model` dict of
local_configs/main_SM3Det.py through the registry (the committed copy sm3det_amd/configs/baseline_configs.json and, when
/root/reference exists, `Config.fromfile` of the file itself); (ii) the composition of `forward_train` (gather by source,
ONE backbone call on the concatenated modalities, split, neck x3 with the per-modality start_level, three heads, loss-dict
keys) PINNED on the reference's own class run live over the same recording stub children (oracle/ref_detector.py)."""
import copy
import json
import os

import pytest
import torch
import torch.nn as nn

from sm3det_amd.registry import MODELS

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _model_cfg():
    with open(os.path.join(ROOT, 'sm3det_amd', 'configs', 'baseline_configs.json')) as f:
        m = json.load(f)['main_SM3Det']['model']
    m = copy.deepcopy(m)
    m['backbone'].pop('init_cfg', None)  # no checkpoint file here
    return m


def test_builds_from_the_config_dict_by_type_string():
    from sm3det_amd import detector  # noqa: F401
    det = MODELS.build(_model_cfg())
    assert type(det).__name__ == 'TriSourceDetector'
    assert type(det.backbone).__name__ == 'ConvNeXt_moe_MultiInput' and type(det.neck).__name__ == 'MultitaskFPN'
    assert type(det.sar_bbox_head).__name__ == 'GFLHead' and type(det.rgb_rpn_head).__name__ == 'OrientedRPNHead'
    assert type(det.ifr_roi_head).__name__ == 'OrientedStandardRoIHead'
    assert det.sar_bbox_head.train_cfg['assigner']['type'] == 'ATSSAssigner'      # sar_train_cfg injected (:101-104)
    assert det.rgb_rpn_head.train_cfg is not None and det.rgb_roi_head.train_cfg is not None
    n = sum(p.numel() for p in det.parameters())
    assert abs(n / 1e6 - 178.17) < 0.05, n  # ~178 M parameters for config #2
    heads = sorted({k.split('.')[0] for k in det.state_dict()})
    assert heads == ['backbone', 'ifr_roi_head', 'ifr_rpn_head', 'neck', 'rgb_roi_head', 'rgb_rpn_head', 'sar_bbox_head']


@pytest.mark.skipif(not os.path.exists('/root/reference/local_configs/main_SM3Det.py'), reason='needs /root/reference')
def test_builds_from_the_reference_config_file_unchanged():
    from sm3det_amd import detector  # noqa: F401
    from sm3det_amd.config import Config
    cfg = Config.fromfile('/root/reference/local_configs/main_SM3Det.py')
    model = copy.deepcopy(cfg.model)
    model['backbone'].pop('init_cfg')  # the ImageNet checkpoint path of the authors' machine
    det = MODELS.build(model)
    assert type(det).__name__ == 'TriSourceDetector' and cfg.model['type'] == 'TriSourceDetector'


# ---- composition pinned on the reference class over recording stubs -----------------------------------------------
CALLS = []


class _StubBackbone(nn.Module):
    def __init__(self, **kw):
        super().__init__()
        self.w = nn.Parameter(torch.tensor(1.5))

    def forward(self, x, datasets):
        CALLS.append(('backbone', [tuple(t.shape) for t in x] if isinstance(x, (list, tuple)) else tuple(x.shape), list(datasets)))
        if isinstance(x, (list, tuple)):
            x = torch.cat(list(x), 0)
        feats = tuple(torch.nn.functional.avg_pool2d(x, 4 * 2 ** i) * self.w * (i + 1) for i in range(4))
        return feats, (self.w * 0.25).sum()


class _StubNeck(nn.Module):
    def __init__(self, **kw):
        super().__init__()

    def forward(self, inputs, start_level=None, add_extra_convs=None):
        CALLS.append(('neck', [tuple(t.shape) for t in inputs], start_level, add_extra_convs))
        s = 0 if start_level is None else start_level
        return tuple(t * 2.0 for t in inputs[s:])


class _StubDense(nn.Module):  # GFL head / RPN head
    def __init__(self, tag='', train_cfg=None, test_cfg=None, **kw):
        super().__init__()
        self.tag, self.train_cfg, self.test_cfg = tag, train_cfg, test_cfg

    def forward_train(self, x, img_metas, gt_bboxes, gt_labels=None, gt_bboxes_ignore=None, proposal_cfg=None, **kw):
        CALLS.append((self.tag, [tuple(t.shape) for t in x], len(img_metas), [tuple(g.shape) for g in gt_bboxes],
                      None if gt_labels is None else [tuple(g.shape) for g in gt_labels], gt_bboxes_ignore,
                      None if proposal_cfg is None else dict(proposal_cfg)))
        v = sum(t.sum() for t in x) * 1e-3
        if self.tag == 'sar':
            return dict(loss_cls=[v, v * 2], loss_bbox=[v * 3], loss_dfl=[v * 4])
        return dict(loss_rpn_cls=[v], loss_rpn_bbox=[v * 2]), [torch.ones(5, 6) * len(img_metas)]


class _StubRoI(nn.Module):
    def __init__(self, tag='', train_cfg=None, test_cfg=None, **kw):
        super().__init__()
        self.tag, self.train_cfg, self.test_cfg = tag, train_cfg, test_cfg

    def forward_train(self, x, img_metas, proposal_list, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_masks=None, **kw):
        CALLS.append((self.tag, [tuple(t.shape) for t in x], len(img_metas), [tuple(p.shape) for p in proposal_list],
                      [tuple(g.shape) for g in gt_bboxes], [tuple(g.shape) for g in gt_labels]))
        v = sum(t.sum() for t in x) * 1e-3 + sum(p.sum() for p in proposal_list)
        return dict(loss_cls=v, loss_bbox=v * 0.5, acc=v * 0 + 87.5)


def _stub_registry():
    from sm3det_amd.registry import Registry
    from sm3det_amd import detector
    r = Registry('stubs')
    for n, c in (('SB', _StubBackbone), ('SN', _StubNeck), ('SD', _StubDense), ('SR', _StubRoI)):
        r.register_module(name=n, module=c)
    r.register_module(name='TriSourceDetector', module=detector.TriSourceDetector)
    return r


def _stub_model_cfg():
    from sm3det_amd.config import ConfigDict
    tr = lambda: ConfigDict(rpn=ConfigDict(a=1), rpn_proposal=ConfigDict(nms_pre=2000, max_per_img=2000), rcnn=ConfigDict(b=2))  # noqa: E731
    te = lambda: ConfigDict(rpn=ConfigDict(nms_pre=1000), rcnn=ConfigDict(score_thr=0.05))  # noqa: E731
    return ConfigDict(backbone=ConfigDict(type='SB'), neck=ConfigDict(type='SN'),
                      rgb_rpn_head=ConfigDict(type='SD', tag='rgb_rpn'), rgb_roi_head=ConfigDict(type='SR', tag='rgb_roi'),
                      rgb_train_cfg=tr(), rgb_test_cfg=te(),
                      ifr_rpn_head=ConfigDict(type='SD', tag='ifr_rpn'), ifr_roi_head=ConfigDict(type='SR', tag='ifr_roi'),
                      ifr_train_cfg=tr(), ifr_test_cfg=te(),
                      sar_bbox_head=ConfigDict(type='SD', tag='sar'), sar_train_cfg=ConfigDict(assigner=1), sar_test_cfg=ConfigDict(x=2))


def _batch(mix, seed=0):
    g = torch.Generator().manual_seed(seed)
    img, metas, gtb, gtl = [], [], [], []
    for i, src in enumerate(mix):
        n = 2 + i
        img.append({src: torch.randn(3, 64, 64, generator=g)})
        metas.append({src: dict(img_shape=(64, 64, 3), pad_shape=(64, 64, 3), idx=i)})
        gtb.append({src: torch.rand(n, 4 if src == 'sar' else 5, generator=g)})
        gtl.append({src: torch.randint(0, 26, (n,), generator=g)})
    return img, metas, gtb, gtl


@pytest.mark.skipif(not os.path.exists('/root/reference/mmrotate'), reason='needs /root/reference')
@pytest.mark.parametrize('mix', [['sar', 'sar', 'rgb', 'ifr'], ['rgb', 'sar', 'ifr', 'sar', 'rgb'], ['sar', 'rgb'], ['ifr', 'ifr']])
def test_forward_train_composition_equals_the_reference_class_run_live(mix, monkeypatch):
    from oracle import ref_detector
    from sm3det_amd import detector, h2d
    reg = _stub_registry()
    ref_mod = ref_detector.load(reg)
    monkeypatch.setattr(torch.Tensor, 'cuda', lambda self, *a, **k: self)          # the reference hard-codes .cuda() (:202-204)
    monkeypatch.setattr(detector, 'MODELS', reg)                                    # the product builds its children from the stubs
    monkeypatch.setattr(h2d, 'gather_dict_values', lambda data, ds, ignore_tensor=False, uploader=None: _cpu_gather(data, ds, ignore_tensor))
    monkeypatch.setattr(h2d, 'PinnedUploader', lambda *a, **k: None)
    ref = ref_mod.TriSourceDetector(**copy.deepcopy(_stub_model_cfg()))
    mine = detector.TriSourceDetector(**copy.deepcopy(_stub_model_cfg()))
    assert ref.rgb_rpn_head.train_cfg == mine.rgb_rpn_head.train_cfg and ref.ifr_roi_head.test_cfg == mine.ifr_roi_head.test_cfg
    assert ref.sar_bbox_head.train_cfg == mine.sar_bbox_head.train_cfg
    batch = _batch(mix)
    CALLS.clear()
    out_ref = ref.forward_train(*copy.deepcopy(batch))
    calls_ref = list(CALLS)
    CALLS.clear()
    out_mine = mine.forward_train(*copy.deepcopy(batch))
    calls_mine = list(CALLS)
    # the one deliberate difference: the reference runs the neck on the EMPTY feature maps of a source without images
    # this step (its outputs are never used); the product skips that call (a zero-row launch), nothing else differs
    calls_ref = [c for c in calls_ref if not (c[0] == 'neck' and c[1][0][0] == 0)]
    assert calls_ref == calls_mine
    assert list(out_ref.keys()) == list(out_mine.keys())
    for k in out_ref:
        a, b = out_ref[k], out_mine[k]
        if isinstance(a, (list, tuple)):
            assert len(a) == len(b) and all(torch.equal(x, y) for x, y in zip(a, b)), k
        else:
            assert torch.equal(a, b), k
    expected = {'gate_loss'}
    if 'sar' in mix:
        expected |= {'sar_loss_cls', 'sar_loss_bbox', 'sar_loss_dfl'}
    for m in ('rgb', 'ifr'):
        if m in mix:
            expected |= {f'{m}_loss_rpn_cls', f'{m}_loss_rpn_bbox', f'{m}_loss_cls', f'{m}_loss_bbox', f'{m}_acc'}
    assert set(out_mine) == expected
    total, logs = detector.TriSourceDetector.parse_losses(out_mine)
    assert torch.isfinite(total) and all('acc' in k or 'loss' in k for k in logs)


def _cpu_gather(data, datasets, ignore_tensor):
    from sm3det_amd.h2d import collect_by_source
    g = collect_by_source(data, datasets)
    for ns in datasets:
        if g[ns] and isinstance(g[ns][0], torch.Tensor) and not ignore_tensor:
            g[ns] = torch.stack(g[ns])
    return g


@pytest.mark.skipif(not os.path.exists('/root/reference/mmrotate'), reason='needs /root/reference')
def test_multi_task_reweighting_branches_equal_the_reference(monkeypatch):
    from oracle import ref_detector
    from sm3det_amd import detector, h2d
    reg = _stub_registry()
    ref_mod = ref_detector.load(reg)
    monkeypatch.setattr(torch.Tensor, 'cuda', lambda self, *a, **k: self)
    monkeypatch.setattr(detector, 'MODELS', reg)
    monkeypatch.setattr(h2d, 'gather_dict_values', lambda data, ds, ignore_tensor=False, uploader=None: _cpu_gather(data, ds, ignore_tensor))
    monkeypatch.setattr(h2d, 'PinnedUploader', lambda *a, **k: None)
    keys = ['sar_loss_cls', 'rgb_loss_cls', 'ifr_loss_bbox']
    cfg = _stub_model_cfg()
    ref = ref_mod.TriSourceDetector(**copy.deepcopy(cfg), multi_tasks_reweight='uncertainty', reweight_losses=keys)
    mine = detector.TriSourceDetector(**copy.deepcopy(cfg), multi_tasks_reweight='uncertainty', reweight_losses=keys)
    batch = _batch(['sar', 'rgb', 'ifr'])
    a, b = ref.forward_train(*copy.deepcopy(batch)), mine.forward_train(*copy.deepcopy(batch))
    assert list(a) == list(b) and torch.allclose(a['reweighted_total_losses'], b['reweighted_total_losses'])
