"""GPU tier: one TRAINING STEP of the whole detector of local_configs/main_SM3Det.py -- `MODELS.build(cfg.model)` ->
`TriSourceDetector.forward_train` on the config's native modality mix (2 SAR + 1 RGB + 1 IR samples) -> one loss dict with
the reference's keys -> backward: every parameter receives a finite gradient.  The SAR branch's losses are re-derived on
the CPU from the head's own outputs with the oracle's indexing-form restatement (oracle/gfl_oracle.py)."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from tests import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RES = 512


def _model_cfg():
    with open(os.path.join(ROOT, 'sm3det_amd', 'configs', 'baseline_configs.json')) as f:
        m = copy.deepcopy(json.load(f)['main_SM3Det']['model'])
    m['backbone'].pop('init_cfg', None)
    return m


def _batch(mix, res=RES, ngt=8):
    g = torch.Generator().manual_seed(7)
    img, metas, gtb, gtl = [], [], [], []
    for i, src in enumerate(mix):
        img.append({src: torch.randn(3, res, res, generator=g)})
        metas.append({src: dict(img_shape=(res, res, 3), pad_shape=(res, res, 3), scale_factor=1.0, flip=False)})
        if src == 'sar':
            b = torch.from_numpy(np.ascontiguousarray(synth.hboxes(ngt, 60 + i, extent=float(res))))
        else:
            b = torch.from_numpy(np.ascontiguousarray(synth.rotated_boxes(ngt, 70 + i))) * torch.tensor([res / 1024.0] * 4 + [1.0])
        gtb.append({src: b.float()})
        gtl.append({src: torch.randint(0, 26, (ngt,), generator=g)})
    return img, metas, gtb, gtl


def test_full_detector_training_step_on_the_native_modality_mix():
    from oracle import gfl_oracle as GO
    from sm3det_amd import detector  # noqa: F401
    from sm3det_amd.registry import MODELS
    torch.manual_seed(0)
    det = MODELS.build(_model_cfg()).cuda().train()
    assert type(det).__name__ == 'TriSourceDetector'
    for h in (det.rgb_rpn_head, det.rgb_roi_head, det.ifr_rpn_head, det.ifr_roi_head):
        h.init_weights()
    with torch.no_grad():  # layer scale 1e-6 hides the FFN / MoE branch numerically: O(1) like a trained net
        for n, p in det.backbone.named_parameters():
            if n.endswith('gamma'):
                p.fill_(1.0)
    captured = {}
    det.sar_bbox_head.register_forward_hook(lambda m, i, o: captured.update(out=o))
    mix = ['sar', 'sar', 'rgb', 'ifr']  # source_ratio [2, 1, 1] of the config
    img, metas, gtb, gtl = _batch(mix)
    losses = det.forward_train(img, metas, gtb, gtl)
    expected = {'gate_loss', 'sar_loss_cls', 'sar_loss_bbox', 'sar_loss_dfl'}
    for m in ('rgb', 'ifr'):
        expected |= {f'{m}_loss_rpn_cls', f'{m}_loss_rpn_bbox', f'{m}_loss_cls', f'{m}_loss_bbox', f'{m}_acc'}
    assert set(losses) == expected
    total, logs = det.parse_losses(losses)
    assert bool(torch.isfinite(total)) and all(bool(torch.isfinite(v)) for v in logs.values()), {k: float(v) for k, v in logs.items()}
    assert det.source_ratio == [2, 1, 1]
    # SAR branch against the oracle's indexing-form restatement, from the head's own outputs
    cls_scores, bbox_preds = captured['out']
    assert cls_scores[0].shape[0] == 2 and len(cls_scores) == 5 and cls_scores[0].shape[-1] == RES // 8
    sizes = [tuple(c.shape[-2:]) for c in cls_scores]
    from sm3det_amd.rpn_head import grid_anchors
    lvl = grid_anchors(sizes, [8, 16, 32, 64, 128], [8], [1.0], device='cpu')
    sar_b = [d['sar'].float() for d in gtb if 'sar' in d]
    sar_l = [d['sar'] for d in gtl if 'sar' in d]
    exp = GO.gfl_loss([c.detach().float().cpu() for c in cls_scores], [b.detach().float().cpu() for b in bbox_preds], lvl,
                      [8, 16, 32, 64, 128], sar_b, sar_l, 26)
    for key in ('loss_cls', 'loss_bbox', 'loss_dfl'):
        a = float(sum(losses['sar_' + key]))
        b = float(sum(exp[key]))
        assert abs(a - b) <= 2e-4 * max(1.0, abs(b)), (key, a, b)
    total.backward()
    torch.cuda.synchronize()
    missing = [n for n, p in det.named_parameters() if p.requires_grad and p.grad is None]
    assert not missing, missing[:8]
    bad = [n for n, p in det.named_parameters() if p.grad is not None and not bool(torch.isfinite(p.grad).all())]
    assert not bad, bad[:8]
    # each branch's parameters got a gradient that is not identically zero
    for pre in ('backbone.stages.3', 'neck.', 'sar_bbox_head.gfl_cls', 'rgb_rpn_head.', 'rgb_roi_head.bbox_head.fc_cls',
                'ifr_rpn_head.', 'ifr_roi_head.bbox_head.fc_reg'):
        gs = [float(p.grad.abs().sum()) for n, p in det.named_parameters() if n.startswith(pre)]
        assert gs and sum(gs) > 0, pre


def test_detector_skips_the_branch_of_an_absent_source_and_sar_inference_runs():
    from sm3det_amd import detector  # noqa: F401
    from sm3det_amd.registry import MODELS
    torch.manual_seed(1)
    det = MODELS.build(_model_cfg()).cuda().train()
    img, metas, gtb, gtl = _batch(['sar', 'rgb'], res=256, ngt=4)
    losses = det.forward_train(img, metas, gtb, gtl)
    assert not any(k.startswith('ifr_') for k in losses) and 'rgb_loss_cls' in losses and 'sar_loss_dfl' in losses
    det.eval()
    x = torch.randn(1, 3, 256, 256, device='cuda')
    import numpy as np
    meta = [dict(img_shape=(256, 256, 3), scale_factor=np.ones(4, np.float32))]
    # the reference's result type (trisource_H1stage_R2stage_detector.py:371-400): per image a list over the classes
    res = det.simple_test(x, meta, [['sar']])
    assert len(res) == 1 and len(res[0]) == det.sar_bbox_head.num_classes
    assert all(isinstance(a, np.ndarray) and a.ndim == 2 and a.shape[1] == 5 for a in res[0])
    assert sum(a.shape[0] for a in res[0]) <= 100
    for sub in ('rgb', 'ifr'):  # two-stage branches: RPN proposals -> RoI head -> multiclass rotated NMS
        res = det.simple_test(x, meta, [[sub]], rescale=True)
        roi = getattr(det, f'{sub}_roi_head')
        assert len(res) == 1 and len(res[0]) == roi.bbox_head.num_classes
        assert all(isinstance(a, np.ndarray) and a.dtype == np.float32 and a.ndim == 2 and a.shape[1] == 6 for a in res[0])
        assert sum(a.shape[0] for a in res[0]) <= 2000
        dets = np.concatenate(res[0])
        assert np.isfinite(dets).all() and (dets[:, 5] > 0.05).all()


def test_rgb_branch_losses_inside_the_detector_equal_the_pinned_oracles():
    """Detector-level numeric check of the two-stage (RGB) branch: inside `TriSourceDetector.forward_train` the Oriented-RPN
    and RoI-head losses are re-derived on the CPU by the oracles that the reference's own head classes pin
    (oracle/loss_oracle.py via tests/test_oracle_losses.py: `_get_targets_single` / `get_targets` / `loss` run live) -- from
    the tensors the detector actually handed to the heads (FPN outputs of the MoE backbone, the gts of the RGB sample) and
    the anchors / RoIs its samplers actually drew.  Values within 1e-4."""
    from oracle import loss_oracle as LO
    from sm3det_amd import detector  # noqa: F401
    from sm3det_amd.registry import MODELS
    from tests import losses_common as LC
    torch.manual_seed(0)
    det = MODELS.build(_model_cfg()).cuda().train()
    for h in (det.rgb_rpn_head, det.rgb_roi_head, det.ifr_rpn_head, det.ifr_roi_head):
        h.init_weights()
    with torch.no_grad():
        for n, p in det.backbone.named_parameters():
            if n.endswith('gamma'):
                p.fill_(1.0)
    cap = {}
    rpn, roi = det.rgb_rpn_head, det.rgb_roi_head
    rpn_loss, roi_train = rpn.loss, roi.forward_train

    def loss_wrap(cls_scores, bbox_preds, gt_bboxes, img_metas, gt_bboxes_ignore=None, **kw):
        losses, smp = rpn_loss(cls_scores, bbox_preds, gt_bboxes, img_metas, gt_bboxes_ignore, return_samples=True, **kw)
        cap['rpn'] = dict(cls=[c.detach().float().cpu() for c in cls_scores], reg=[r.detach().float().cpu() for r in bbox_preds],
                          gts=[g.detach().float().cpu() for g in gt_bboxes], smp={k: v.detach().cpu() for k, v in smp.items()},
                          losses=losses)
        return losses

    def roi_wrap(x, img_metas, proposal_list, gt_bboxes, gt_labels, *a, **kw):
        losses, smp = roi_train(x, img_metas, proposal_list, gt_bboxes, gt_labels, *a, return_samples=True, **kw)
        with torch.no_grad():
            res = roi._bbox_forward(x, smp['rois'])
        cap['roi'] = dict(smp={k: v.detach().cpu() for k, v in smp.items() if torch.is_tensor(v)}, losses=losses,
                          cls=res['cls_score'].float().cpu(), reg=res['bbox_pred'].float().cpu())
        return losses

    rpn.loss, roi.forward_train = loss_wrap, roi_wrap
    img, metas, gtb, gtl = _batch(['sar', 'sar', 'rgb', 'ifr'])
    losses = det.forward_train(img, metas, gtb, gtl)
    # ---- Oriented RPN: oracle on the captured head outputs, the captured anchors and the sampler's picks
    r = cap['rpn']
    smp = r['smp']
    sizes = [tuple(c.shape[-2:]) for c in r['cls']]
    counts = [h * w * 3 for h, w in sizes]
    anchors = list(smp['anchors'].split(counts))
    idx, is_pos, valid = smp['idx'], smp['is_pos'].bool(), smp['valid'].bool()
    pos = [idx[i][is_pos[i] & valid[i]] for i in range(idx.shape[0])]
    neg = [idx[i][(~is_pos[i]) & valid[i]] for i in range(idx.shape[0])]
    assert all(p.numel() > 0 for p in pos) and all(n.numel() > 0 for n in neg)
    oc, ob = LO.rpn_loss(r['cls'], r['reg'], anchors, smp['inside'].bool(), r['gts'], pos, neg, LC.RPN_MEANS, LC.RPN_STDS,
                         beta=1.0 / 9.0, assign_cfg=LC.RPN_ASSIGN)
    for key, exp in (('rgb_loss_rpn_cls', oc), ('rgb_loss_rpn_bbox', ob)):
        a, b = float(sum(losses[key])), float(sum(exp))
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (key, a, b)
    # ---- RoI head: oracle on the head's scores / deltas for the RoIs its sampler drew
    q = cap['roi']
    s_ = q['smp']
    C = roi.bbox_head.num_classes
    lab, val = s_['labels'], s_['valid'].bool()
    rois, gts_s = s_['rois'], s_['gts']
    assert bool(val.any()) and int((rois[:, 0] != 0).sum()) == 0  # one RGB image in the batch
    p_ = (lab < C) & val
    n_ = (lab >= C) & val
    exp = LO.rcnn_loss(q['cls'][val], q['reg'][val], [rois[p_][:, 1:]], [rois[n_][:, 1:]], [gts_s[p_]], [lab[p_]], C,
                       LC.RCNN_MEANS, LC.RCNN_STDS)
    for key in ('loss_cls', 'loss_bbox', 'acc'):
        a, b = float(losses['rgb_' + key]), float(exp[key])
        assert abs(a - b) <= 1e-4 * max(1.0, abs(b)), (key, a, b)
    assert float(losses['rgb_loss_bbox']) > 0
