"""``TriSourceDetector`` -- the detector composition of ``local_configs/main_SM3Det.py`` on the MI355X pieces.

Mirror of ``mmrotate/models/detectors/trisource_H1stage_R2stage_detector.py:27-369``: one MoE backbone call on the
concatenated modalities, split by source, one shared ``MultitaskFPN`` applied three times (SAR: ``start_level=1,
add_extra_convs='on_output'``; RGB / IR: as configured), a one-stage ``GFLHead`` for the SAR images (horizontal boxes) and
an Oriented-RCNN pair (``OrientedRPNHead`` + ``OrientedStandardRoIHead``) each for the RGB and the IR images, one loss dict
with the reference's keys (``gate_loss``, ``sar_loss_{cls,bbox,dfl}``, ``{rgb,ifr}_loss_rpn_{cls,bbox}``,
``{rgb,ifr}_{loss_cls,loss_bbox,acc}``).  Registered by ``type`` string so that ``MODELS.build(Config.fromfile(
'local_configs/main_SM3Det.py').model)`` returns it with the config dict unchanged.

What runs where: backbone / neck / RPN + RoI heads / GFL towers and the two-stage losses are the HIP kernels of this
package (hot paths (a), (b) and SURVEY 8(f) rows 2-3); the SAR branch's ATSS assignment and QFL / DFL / GIoU losses are
plain PyTorch over fixed-shape masked tensors (mmdet code the reference does not vendor: restated, parity unpinned --
``sm3det_amd/gfl_losses.py``).  ``forward_train`` performs no device->host synchronisation for a fixed number of ground
truths per image, so a whole training step can be captured into hipGraphs (``bench.py``: ``full_model_imgs_per_sec``).
The multi-task reweighting branches (``multi_tasks_reweight='uncertainty' | 'dwa'``, :306-338) are restated too; no
SM3Det config of BASELINE.json enables them.
"""
import os

import torch
import torch.nn as nn

from . import backbone_ops, h2d
from .registry import MODELS

_PAIR_IN_DETECTOR = int(os.environ.get('SM3_PAIR_DGRAD_DETECTOR', '0'))


@MODELS.register_module()
class TriSourceDetector(nn.Module):
    train_datasets = ['sar', 'rgb', 'ifr']

    def __init__(self, backbone, neck=None, rgb_rpn_head=None, rgb_roi_head=None, rgb_train_cfg=None, rgb_test_cfg=None,
                 ifr_rpn_head=None, ifr_roi_head=None, ifr_train_cfg=None, ifr_test_cfg=None, sar_bbox_head=None,
                 sar_train_cfg=None, sar_test_cfg=None, multi_tasks_reweight=None, reweight_losses=None, train_cfg=None,
                 test_cfg=None, pretrained=None, init_cfg=None):
        super().__init__()
        from . import convnext_moe, fpn, gfl_head, roi_head, rpn_head  # noqa: F401  (registration side effects)
        self.init_cfg = init_cfg
        backbone = dict(backbone)
        if pretrained:  # :55-58 (deprecated spelling of init_cfg=dict(type='Pretrained', checkpoint=...))
            backbone['init_cfg'] = dict(type='Pretrained', checkpoint=pretrained)
        self.backbone = MODELS.build(backbone)
        self.train_datasets = ['sar', 'rgb', 'ifr']
        if neck is not None:
            self.neck = MODELS.build(neck)
        for mod, rpn, roi, tr, te in (('rgb', rgb_rpn_head, rgb_roi_head, rgb_train_cfg, rgb_test_cfg),
                                      ('ifr', ifr_rpn_head, ifr_roi_head, ifr_train_cfg, ifr_test_cfg)):
            if rpn is not None:  # :64-68 / :83-87
                setattr(self, f'{mod}_rpn_head', MODELS.build(dict(
                    rpn, train_cfg=(tr or {}).get('rpn') if tr is not None else None, test_cfg=(te or {}).get('rpn'))))
            if roi is not None:  # :70-77 / :89-96
                setattr(self, f'{mod}_roi_head', MODELS.build(dict(
                    roi, train_cfg=(tr or {}).get('rcnn') if tr is not None else None, test_cfg=(te or {}).get('rcnn'))))
        self.rgb_train_cfg, self.rgb_test_cfg = rgb_train_cfg, rgb_test_cfg
        self.ifr_train_cfg, self.ifr_test_cfg = ifr_train_cfg, ifr_test_cfg
        # :101-104 (the reference builds the SAR head unconditionally)
        self.sar_bbox_head = MODELS.build(dict(sar_bbox_head, train_cfg=sar_train_cfg, test_cfg=sar_test_cfg))
        self.sar_train_cfg, self.sar_test_cfg = sar_train_cfg, sar_test_cfg
        self.multi_tasks_reweight, self.reweight_losses = multi_tasks_reweight, reweight_losses
        if multi_tasks_reweight == 'uncertainty':  # :111-113
            self.mtl_sigma = nn.Parameter(torch.ones(len(reweight_losses)))
        elif multi_tasks_reweight == 'dwa':
            self.T = 3
            self.history_loss = None
        self.source_ratio = None
        self._uploader = None

    # ---- the reference's properties -------------------------------------------------------------------------------
    @property
    def with_neck(self):
        return getattr(self, 'neck', None) is not None

    @property
    def with_rgb_rpn(self):
        return getattr(self, 'rgb_rpn_head', None) is not None

    @property
    def with_rgb_roi_head(self):
        return getattr(self, 'rgb_roi_head', None) is not None

    @property
    def with_ifr_rpn(self):
        return getattr(self, 'ifr_rpn_head', None) is not None

    @property
    def with_ifr_roi_head(self):
        return getattr(self, 'ifr_roi_head', None) is not None

    def init_weights(self):
        """BaseModule.init_weights: every child that defines one (the backbone loads its `Pretrained` checkpoint here)"""
        for m in self.children():
            if hasattr(m, 'init_weights'):
                m.init_weights()

    # ---- features ------------------------------------------------------------------------------------------------
    def extract_feat(self, batch_inputs, datasets, is_train=False):
        """:141-173.  batch_inputs: list of per-source image stacks (train) or one tensor (test)."""
        # (SM3_PAIR_DGRAD_DETECTOR: concurrent backward partners inside the backbone, off in the whole-detector step --
        # measured slower under one large hipGraph, see backbone_ops.pairing)
        with backbone_ops.pairing(_PAIR_IN_DETECTOR):
            x = self.backbone(batch_inputs, datasets)
        loss = None
        if isinstance(x, tuple) and len(x) == 2 and not torch.is_tensor(x[0]):
            x, loss = x
        if self.with_neck:
            if len(datasets) > 1:
                assert is_train
                sar_x, rgb_x, ifr_x = self.split_batch(x)
                # a source without images this step yields empty feature maps; its branch is skipped below
                sar_x = self.neck(sar_x, start_level=1, add_extra_convs='on_output') if sar_x[0].shape[0] else None
                rgb_x = self.neck(rgb_x) if rgb_x[0].shape[0] else None
                ifr_x = self.neck(ifr_x) if ifr_x[0].shape[0] else None
                x = (sar_x, rgb_x, ifr_x)
            else:
                assert not is_train
                if datasets[0] == 'sar':
                    x = self.neck(x, start_level=1, add_extra_convs='on_output')
                elif datasets[0] in ('rgb', 'ifr'):
                    x = self.neck(x)
                else:
                    raise AssertionError('Invalid dataset')
        if is_train:
            return x, loss
        return x, None

    def split_batch(self, x, is_list=False):
        """:175-187"""
        if is_list:
            out, start = [], 0
            for n in self.source_ratio:
                out.append(x[start:start + n])
                start += n
            return out
        slices = [torch.split(x_, self.source_ratio, dim=0) for x_ in x]
        return tuple(map(list, zip(*slices)))

    def gather_dict_values(self, data, ignore_tensor=False):
        """:190-206 through the pinned, non-blocking uploader (sm3det_amd/h2d.py)"""
        if self._uploader is None:
            self._uploader = h2d.PinnedUploader()
        return h2d.gather_dict_values(data, self.train_datasets, ignore_tensor, self._uploader)

    # ---- training ------------------------------------------------------------------------------------------------
    def forward_train(self, img, img_metas, gt_bboxes, gt_labels, gt_bboxes_ignore=None, gt_masks=None, proposals=None,
                      **kwargs):
        """:235-369.  Every argument is a list with one dict per sample, keyed by the sample's source ('sar' | 'rgb' |
        'ifr'), exactly as the reference's multi-source collate hands them over."""
        assert gt_bboxes_ignore is None
        img = self.gather_dict_values(img)
        img_metas = self.gather_dict_values(img_metas)
        gt_bboxes = self.gather_dict_values(gt_bboxes, ignore_tensor=True)
        gt_labels = self.gather_dict_values(gt_labels, ignore_tensor=True)
        return self.forward_train_gathered(img, img_metas, gt_bboxes, gt_labels, proposals=proposals, **kwargs)

    def forward_train_gathered(self, img, img_metas, gt_bboxes, gt_labels, proposals=None, **kwargs):
        """the body of forward_train on per-source dicts that are already on the device (what a captured step replays)"""
        self.source_ratio = [len(gt_labels['sar']), len(gt_labels['rgb']), len(gt_labels['ifr'])]
        batch_inputs = [img[s] for s in self.train_datasets if len(img[s]) > 0]
        x, gate_loss = self.extract_feat(batch_inputs, self.train_datasets, is_train=True)
        losses = dict()
        if gate_loss is not None:
            losses['gate_loss'] = gate_loss
        sar_x, rgb_x, ifr_x = x
        if len(gt_labels['sar']) > 0:
            shape = tuple(img['sar'][0].shape[-2:])
            for m in img_metas['sar']:
                m['batch_input_shape'] = shape
            sl = self.sar_bbox_head.forward_train(sar_x, img_metas['sar'], gt_bboxes['sar'], gt_labels['sar'], None)
            losses.update({'sar_' + k: v for k, v in sl.items()})
        for mod, feats in (('rgb', rgb_x), ('ifr', ifr_x)):
            if len(gt_labels[mod]) == 0:
                continue
            tr, te = getattr(self, f'{mod}_train_cfg'), getattr(self, f'{mod}_test_cfg')
            if getattr(self, f'with_{mod}_rpn'):
                proposal_cfg = (tr or {}).get('rpn_proposal', (te or {}).get('rpn'))
                rpn_losses, proposal_list = getattr(self, f'{mod}_rpn_head').forward_train(
                    feats, img_metas[mod], gt_bboxes[mod], gt_labels=None, gt_bboxes_ignore=None,
                    proposal_cfg=proposal_cfg, **kwargs)
                losses.update({f'{mod}_' + k: v for k, v in rpn_losses.items()})
            else:
                proposal_list = proposals
            roi_losses = getattr(self, f'{mod}_roi_head').forward_train(feats, img_metas[mod], proposal_list,
                                                                        gt_bboxes[mod], gt_labels[mod], None, None, **kwargs)
            losses.update({f'{mod}_' + k: v for k, v in roi_losses.items()})
        if self.multi_tasks_reweight is None:
            return losses
        # :306-338
        out, cur = {}, []
        for k, v in losses.items():
            if k not in self.reweight_losses:
                out[k] = v
                continue
            cur.append(sum(v) if isinstance(v, (list, tuple)) else v)
        cur = torch.stack(cur)
        if self.multi_tasks_reweight == 'uncertainty':
            total = 0
            for i, l in enumerate(cur):
                total = total + 0.5 / (self.mtl_sigma[i] ** 2) * l + torch.log(1 + self.mtl_sigma[i] ** 2)
        elif self.multi_tasks_reweight == 'dwa':
            if self.history_loss is not None:
                w = len(self.reweight_losses) * torch.softmax(cur / self.history_loss / self.T, dim=-1)
            else:
                w = torch.ones_like(cur)
            total = (cur * w).sum()
            self.history_loss = cur.detach()  # kept on the device (the reference round-trips it through numpy)
        else:
            raise NotImplementedError(f'multi_tasks_reweight={self.multi_tasks_reweight!r}')
        out['reweighted_total_losses'] = total
        return out

    @staticmethod
    def parse_losses(losses):
        """mmdet BaseDetector._parse_losses: total = sum of every entry whose key contains 'loss' (lists are summed);
        returns (total, log_vars as device tensors -- no .item(): the caller decides when to synchronise)"""
        log_vars = {}
        for k, v in losses.items():
            if torch.is_tensor(v):
                log_vars[k] = v.mean()
            elif isinstance(v, (list, tuple)):
                log_vars[k] = sum(t.mean() for t in v)
            else:
                raise TypeError(f'{k} is not a tensor or list of tensors')
        total = sum(v for k, v in log_vars.items() if 'loss' in k)
        return total, log_vars

    # ---- inference -----------------------------------------------------------------------------------------------
    def forward_dummy(self, img):
        """:208-233: the backbone runs once per modality (that is how the README's 487 GMACs are counted)"""
        outs = ()
        for mod in ('rgb', 'ifr'):
            x, _ = self.extract_feat(img, [mod])
            outs = outs + (getattr(self, f'{mod}_rpn_head')(x),)
            roi = getattr(self, f'{mod}_roi_head')
            proposals = torch.randn(1000, 5, device=img.device)
            rois = torch.cat([proposals.new_zeros(1000, 1), proposals], 1)
            outs = outs + (roi._bbox_forward(x, rois),)
        x, _ = self.extract_feat(img, ['sar'])
        return outs + (self.sar_bbox_head(x),)

    @torch.no_grad()
    def simple_test(self, img, img_metas, subdataset, proposals=None, rescale=False):
        """:371-400.  Every branch returns the reference's result type: per image a list over the classes of float32
        arrays -- (k, 5) horizontal detections for the SAR branch (``bbox2result`` of ``GFLHead.simple_test``), (k, 6)
        rotated detections (cx, cy, w, h, a, score) for the RGB / IR branches (``OrientedStandardRoIHead.simple_test``:
        RPN proposals -> RoI extractor + Shared2FC -> decode + rescale -> multiclass rotated NMS)."""
        from .post_processing import bbox2result
        assert isinstance(subdataset[0], list) and len(subdataset) == 1
        assert all(s == subdataset[0][0] for s in subdataset[0]), f'Not all elements in subdataset are the same: {subdataset}'
        sub = subdataset[0][0]
        x, _ = self.extract_feat(img, [sub])
        if sub == 'sar':
            results = self.sar_bbox_head.simple_test(x, img_metas, rescale=rescale)
            return [bbox2result(d, l, self.sar_bbox_head.num_classes) for d, l in results]
        if sub in ('rgb', 'ifr'):
            if proposals is None:
                proposals = getattr(self, f'{sub}_rpn_head').simple_test_rpn(x, img_metas)
            return getattr(self, f'{sub}_roi_head').simple_test(x, proposals, img_metas, rescale=rescale)
        raise AssertionError('Invalid dataset')
