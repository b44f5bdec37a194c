"""Import the REFERENCE head classes -- ``mmrotate/models/dense_heads/{rotated_rpn_head,oriented_rpn_head}.py`` and
``mmrotate/models/roi_heads/bbox_heads/rotated_bbox_head.py`` -- unmodified from /root/reference, so that their own
``loss`` / ``get_targets`` / ``_get_targets_single`` / ``loss_single`` code pins oracle/loss_oracle.py.

TEST INFRASTRUCTURE ONLY (CPU tests in the build container; the GPU box has no /root/reference).

Stand-ins registered in ``sys.modules`` for the duration of the import (mmdet / mmcv are not
importable here): ``mmcv.ops.batched_nms`` (never called by the loss path), ``mmcv.runner.{force_fp32, auto_fp16,
BaseModule}`` (identity decorators / nn.Module), ``mmcv.utils.to_2tuple``, ``mmdet.core.{anchor_inside_flags,
images_to_levels, multi_apply, unmap}``, ``mmdet.models.losses.accuracy``, ``mmdet.models.utils.build_linear_layer``,
``mmdet.models.dense_heads.anchor_head.AnchorHead`` (empty nn.Module: every method on the loss path is defined in the
reference files themselves), ``<pkg>.builder.{ROTATED_HEADS, build_loss}`` -- the mmdet functions come from
oracle/loss_oracle.py (restated, unpinned) -- and ``mmrotate.core.{obb2xyxy, build_bbox_coder, multiclass_nms_rotated}``
where obb2xyxy and the coders are the LIVE reference ones (oracle/ref_rpn.py)."""
import importlib.util
import os
import sys

import torch.nn as nn

from oracle import loss_oracle as LO
from oracle import ref_rpn
from oracle.ref_moe import REF_ROOT, _Registry, _mod

_PKG = '_sm3det_ref_pkg_heads'
FILES = {
    f'{_PKG}.dense_heads.rotated_rpn_head': ('mmrotate', 'models', 'dense_heads', 'rotated_rpn_head.py'),
    f'{_PKG}.dense_heads.oriented_rpn_head': ('mmrotate', 'models', 'dense_heads', 'oriented_rpn_head.py'),
    f'{_PKG}.roi_heads.bbox_heads.rotated_bbox_head': ('mmrotate', 'models', 'roi_heads', 'bbox_heads',
                                                       'rotated_bbox_head.py'),
}


def available():
    return ref_rpn.available() and all(os.path.exists(os.path.join(REF_ROOT, *p)) for p in FILES.values())


def load():
    """-> (rotated_rpn_head module, oriented_rpn_head module, rotated_bbox_head module)"""
    if not available():
        raise FileNotFoundError(os.path.join(REF_ROOT, *list(FILES.values())[0]))
    if all(k in sys.modules for k in FILES):
        return tuple(sys.modules[k] for k in FILES)
    T, C, X = ref_rpn.load()
    ident = lambda *a, **k: (lambda f: f)  # noqa: E731

    def build_bbox_coder(cfg):
        cfg = dict(cfg)
        t = cfg.pop('type')
        return {'MidpointOffsetCoder': C.MidpointOffsetCoder, 'DeltaXYWHAOBBoxCoder': X.DeltaXYWHAOBBoxCoder}[t](**cfg)

    class _Base(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()
            self.init_cfg = init_cfg

    def _never(*a, **k):
        raise AssertionError('not on the loss path')

    shims = {
        'mmcv': _mod('mmcv'), 'mmcv.ops': _mod('mmcv.ops', batched_nms=_never),
        'mmcv.runner': _mod('mmcv.runner', force_fp32=ident, auto_fp16=ident, BaseModule=_Base),
        'mmcv.utils': _mod('mmcv.utils', to_2tuple=lambda v: (v, v) if not isinstance(v, (tuple, list)) else tuple(v)),
        'mmdet': _mod('mmdet'),
        'mmdet.core': _mod('mmdet.core', anchor_inside_flags=LO.anchor_inside_flags, images_to_levels=LO.images_to_levels,
                           multi_apply=LO.multi_apply, unmap=LO.unmap),
        'mmdet.models': _mod('mmdet.models'), 'mmdet.models.dense_heads': _mod('mmdet.models.dense_heads'),
        'mmdet.models.dense_heads.anchor_head': _mod('mmdet.models.dense_heads.anchor_head',
                                                     AnchorHead=type('AnchorHead', (nn.Module,), {})),
        'mmdet.models.losses': _mod('mmdet.models.losses', accuracy=LO.accuracy),
        'mmdet.models.utils': _mod('mmdet.models.utils',
                                   build_linear_layer=lambda cfg, *a, **k: nn.Linear(*a, **k)),
        'mmrotate': _mod('mmrotate'),
        'mmrotate.core': _mod('mmrotate.core', obb2xyxy=T.obb2xyxy, build_bbox_coder=build_bbox_coder,
                              multiclass_nms_rotated=_never),
        _PKG: _mod(_PKG, __path__=[]),
        f'{_PKG}.builder': _mod(f'{_PKG}.builder', ROTATED_HEADS=_Registry(), build_loss=LO.build_loss),
        f'{_PKG}.dense_heads': _mod(f'{_PKG}.dense_heads', __path__=[]),
        f'{_PKG}.roi_heads': _mod(f'{_PKG}.roi_heads', __path__=[]),
        f'{_PKG}.roi_heads.bbox_heads': _mod(f'{_PKG}.roi_heads.bbox_heads', __path__=[]),
    }
    saved = {k: sys.modules.get(k) for k in shims}
    sys.modules.update(shims)
    try:
        mods = []
        for name, parts in FILES.items():
            spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, *parts))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[name] = mod
            spec.loader.exec_module(mod)
            mods.append(mod)
    finally:
        for k, v in saved.items():
            if k.startswith(_PKG):
                continue
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return tuple(mods)


class _Cfg(dict):
    __getattr__ = dict.__getitem__


class _FixedSampling:
    """stands in for SamplingResult + sampler: the positives / negatives are the ones the caller chose"""

    def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_gt_inds, gt_labels=None):
        self.pos_inds, self.neg_inds = pos_inds, neg_inds
        self.pos_bboxes, self.neg_bboxes = bboxes[pos_inds], bboxes[neg_inds]
        self.pos_assigned_gt_inds = assign_gt_inds[pos_inds] - 1
        self.pos_gt_bboxes = gt_bboxes[self.pos_assigned_gt_inds] if gt_bboxes.numel() else gt_bboxes.view(-1, 4)
        self.pos_gt_labels = gt_labels[self.pos_assigned_gt_inds] if gt_labels is not None else None


def make_reference_rpn_head(mlvl_anchors, assign_cfg, picks, means, stds, beta, allowed_border=0, pos_weight=-1):
    """A live-reference OrientedRPNHead object with just the attributes its loss path reads (AnchorHead.__init__ is
    mmdet code and is not run): anchors from `mlvl_anchors` (all valid), assigner = oracle/assign_oracle.py's MaxIoU rule
    on the inside anchors, sampler = `picks[i]` = (pos flat indices, neg flat indices) of image i -- translated to the
    compacted inside-anchor numbering the reference works in."""
    import numpy as np
    import torch
    from oracle import assign_oracle
    _, O, _ = load()
    _, C, _ = ref_rpn.load()
    head = O.OrientedRPNHead.__new__(O.OrientedRPNHead)
    nn.Module.__init__(head)
    head.version, head.num_classes, head.cls_out_channels, head.use_sigmoid_cls = 'le90', 1, 1, True
    head.sampling, head.reg_decoded_bbox = True, False
    head.train_cfg = _Cfg(allowed_border=allowed_border, pos_weight=pos_weight)
    head.bbox_coder = C.MidpointOffsetCoder(target_means=means, target_stds=stds, angle_range='le90')
    head.loss_cls = LO.CrossEntropyLoss(use_sigmoid=True, loss_weight=1.0)
    head.loss_bbox = LO.SmoothL1Loss(beta=beta, loss_weight=1.0)
    head.anchor_generator = _Cfg(num_levels=len(mlvl_anchors))
    state = {'img': 0}

    def get_anchors(featmap_sizes, img_metas, device='cpu'):
        n = len(img_metas)
        return ([[a.clone() for a in mlvl_anchors] for _ in range(n)],
                [[torch.ones(a.shape[0], dtype=torch.bool) for a in mlvl_anchors] for _ in range(n)])

    class _Assigner:
        def assign(self, anchors, gt_hbboxes, gt_bboxes_ignore, gt_labels):
            gi, mo, _, _ = assign_oracle.max_iou_assign(anchors.numpy(), gt_hbboxes.numpy(), False, **assign_cfg)
            return _Cfg(gt_inds=torch.from_numpy(gi), max_overlaps=torch.from_numpy(np.asarray(mo)))

    class _Sampler:
        def sample(self, assign_result, anchors, gt_hbboxes):
            i = state['img']
            state['img'] += 1
            flat = torch.cat(mlvl_anchors)
            inside = LO.anchor_inside_flags(flat, torch.ones(flat.shape[0], dtype=torch.bool), (10 ** 9, 10 ** 9), -1)
            inside = state['inside']
            remap = torch.cumsum(inside.long(), 0) - 1  # flat index -> index among the inside anchors
            pos, neg = remap[picks[i][0].long()], remap[picks[i][1].long()]
            gi = assign_result.gt_inds
            assert bool((gi[pos] > 0).all()) and bool((gi[neg] == 0).all()), 'the picks disagree with the assignment'
            return _FixedSampling(pos, neg, anchors, gt_hbboxes, gi)

    head.get_anchors = get_anchors
    head.assigner, head.sampler = _Assigner(), _Sampler()
    head._oracle_state = state
    return head


# ---------------------------------------------------------------------------------------------------------------------
# inference-side classes run LIVE on CPU: OrientedRPNHead._get_bboxes_single with the reference's own `batched_nms`,
# RotatedSingleRoIExtractor with the reference's own RoIAlignRotated wrapper, RotatedShared2FCBBoxHead.forward -- every
# native operator underneath is the reference's CPU C++ compiled by oracle/build_ref.py.  They pin oracle/rpn_oracle.py
# (`rpn_forward_single`, `get_bboxes_single`) and oracle/roi_oracle.py (`extract`, `extract_backward`,
# `shared2fc_forward`), which the GPU tests use as the checker (tests/test_oracle_heads_live.py).
def compiled_reference_ext():
    """an ``mmcv._ext``-shaped module over the reference's CPU operators (oracle/_ref): the pybind shim of build_ref.py
    declares no argument names, so the keyword calls of the python wrappers are mapped to positionals here"""
    import types

    from oracle import build_ref
    ref = build_ref.load_ref()
    m = types.ModuleType('mmcv._ext')
    m.nms = lambda boxes, scores, iou_threshold, offset: ref.nms(boxes, scores, float(iou_threshold), int(offset))
    m.nms_rotated = lambda dets, scores, order, dets_sorted, iou_threshold, multi_label: \
        ref.nms_rotated_cpu(dets, scores, float(iou_threshold))  # pytorch/nms_rotated.cpp: the CPU branch ignores the rest
    m.box_iou_rotated = lambda b1, b2, ious, mode_flag=0, aligned=False: \
        ref.box_iou_rotated(b1, b2, ious, int(mode_flag), bool(aligned))

    def fwd(input, rois, output, pooled_height, pooled_width, spatial_scale, sampling_ratio, aligned, clockwise):
        return ref.roi_align_rotated_forward(input, rois, output, int(pooled_height), int(pooled_width),
                                             float(spatial_scale), int(sampling_ratio), bool(aligned), bool(clockwise))

    def bwd(grad_output, rois, grad_input, pooled_height, pooled_width, spatial_scale, sampling_ratio, aligned,
            clockwise):
        return ref.roi_align_rotated_backward(grad_output, rois, grad_input, int(pooled_height), int(pooled_width),
                                              float(spatial_scale), int(sampling_ratio), bool(aligned), bool(clockwise))
    m.roi_align_rotated_forward, m.roi_align_rotated_backward = fwd, bwd

    def _missing(name):
        def stub(*a, **k):
            raise NotImplementedError(name)
        return stub
    def _getattr(name):  # load_ext asserts that every name of a wrapper file exists; dunders must stay absent (inspect)
        if name.startswith('__'):
            raise AttributeError(name)
        return _missing(name)
    m.__getattr__ = _getattr
    return m


_EXTRACTOR_FILE = ('mmrotate', 'models', 'roi_heads', 'roi_extractors', 'rotate_single_level_roi_extractor.py')
_CONVFC_FILE = ('mmrotate', 'models', 'roi_heads', 'bbox_heads', 'convfc_rbbox_head.py')


def load_inference():
    """-> dict(oriented_rpn_head=<module, its `batched_nms` = the reference's own>, extractor=RotatedSingleRoIExtractor,
    shared2fc=RotatedShared2FCBBoxHead, ops=<the reference's mmcv.ops wrapper modules>)."""
    import torch

    from oracle import ref_mmcv_ops
    _, O, _ = load()
    ops = ref_mmcv_ops.load(ext=compiled_reference_ext())  # reference wrappers over reference C++ (assembles `mmcv*`)
    O.batched_nms = ops['nms'].batched_nms
    name_e, name_c = f'{_PKG}.roi_heads.roi_extractors.rotate_single_level_roi_extractor', \
        f'{_PKG}.roi_heads.bbox_heads.convfc_rbbox_head'
    if name_e in sys.modules and name_c in sys.modules:
        return dict(oriented_rpn_head=O, extractor=sys.modules[name_e].RotatedSingleRoIExtractor,
                    shared2fc=sys.modules[name_c].RotatedShared2FCBBoxHead, ops=ops)
    ident = lambda *a, **k: (lambda f: f)  # noqa: E731

    class _Base(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()
            self.init_cfg = init_cfg

    class BaseRoIExtractor(_Base):  # [memory] mmdet 2.x base_roi_extractor.py: what the subclass's own code relies on
        def __init__(self, roi_layer, out_channels, featmap_strides, init_cfg=None):
            super().__init__(init_cfg)
            self.roi_layers = self.build_roi_layers(roi_layer, featmap_strides)
            self.out_channels, self.featmap_strides, self.fp16_enabled = out_channels, featmap_strides, False

        @property
        def num_inputs(self):
            return len(self.featmap_strides)

    mmcv_ops = _mod('mmcv.ops', RoIAlignRotated=ops['roi_align_rotated'].RoIAlignRotated,
                    RiRoIAlignRotated=type('RiRoIAlignRotated', (), {}), batched_nms=ops['nms'].batched_nms)
    T, C, X = ref_rpn.load()

    def build_bbox_coder(cfg):
        cfg = dict(cfg)
        t = cfg.pop('type')
        return {'MidpointOffsetCoder': C.MidpointOffsetCoder, 'DeltaXYWHAOBBoxCoder': X.DeltaXYWHAOBBoxCoder}[t](**cfg)

    shims = {
        'mmcv': _mod('mmcv', ops=mmcv_ops), 'mmcv.ops': mmcv_ops,
        'mmcv.cnn': _mod('mmcv.cnn', ConvModule=type('ConvModule', (nn.Module,), {})),
        'mmcv.runner': _mod('mmcv.runner', force_fp32=ident, auto_fp16=ident, BaseModule=_Base),
        'mmcv.utils': _mod('mmcv.utils', to_2tuple=lambda v: (v, v) if not isinstance(v, (tuple, list)) else tuple(v)),
        'mmdet': _mod('mmdet'), 'mmdet.core': _mod('mmdet.core', multi_apply=LO.multi_apply),
        'mmdet.models': _mod('mmdet.models'), 'mmdet.models.roi_heads': _mod('mmdet.models.roi_heads'),
        'mmdet.models.roi_heads.roi_extractors': _mod('mmdet.models.roi_heads.roi_extractors'),
        'mmdet.models.roi_heads.roi_extractors.base_roi_extractor':
            _mod('mmdet.models.roi_heads.roi_extractors.base_roi_extractor', BaseRoIExtractor=BaseRoIExtractor),
        'mmdet.models.losses': _mod('mmdet.models.losses', accuracy=LO.accuracy),
        'mmdet.models.utils': _mod('mmdet.models.utils', build_linear_layer=lambda cfg, *a, **k: nn.Linear(*a, **k)),
        # forward() reads these two to pick `.output_size` (mmcv != 1.4.5)
        'mmrotate': _mod('mmrotate', digit_version=lambda v: tuple(int(x) for x in v.split('.')), mmcv_version=(1, 6, 1)),
        'mmrotate.core': _mod('mmrotate.core', build_bbox_coder=build_bbox_coder, multiclass_nms_rotated=None),
        f'{_PKG}.builder': _mod(f'{_PKG}.builder', ROTATED_HEADS=_Registry(), ROTATED_ROI_EXTRACTORS=_Registry(),
                                build_loss=LO.build_loss),
        f'{_PKG}.roi_heads.roi_extractors': _mod(f'{_PKG}.roi_heads.roi_extractors', __path__=[]),
    }
    keep_alive = ('mmrotate',)  # imported lazily inside RotatedSingleRoIExtractor.forward
    saved = {k: sys.modules.get(k) for k in shims}
    sys.modules.update(shims)
    try:
        mods = []
        for name, parts in ((name_e, _EXTRACTOR_FILE), (name_c, _CONVFC_FILE)):
            spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, *parts))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[name] = mod
            spec.loader.exec_module(mod)
            mods.append(mod)
    finally:
        for k, v in saved.items():
            if k.startswith(_PKG) or k in keep_alive:
                continue
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    del torch
    return dict(oriented_rpn_head=O, extractor=mods[0].RotatedSingleRoIExtractor,
                shared2fc=mods[1].RotatedShared2FCBBoxHead, ops=ops)


_ROI_HEAD_FILES = (('mmrotate', 'models', 'roi_heads', 'rotate_standard_roi_head.py'),
                   ('mmrotate', 'models', 'roi_heads', 'oriented_standard_roi_head.py'))


def load_roi_head(assigner_factory, sampler_factory):
    """-> the reference's own ``OrientedStandardRoIHead`` class (oriented_standard_roi_head.py over
    rotate_standard_roi_head.py), importable on CPU: ``build_head`` / ``build_roi_extractor`` resolve to the LIVE reference
    classes of `load_inference`, ``rbbox2roi`` / ``obb2xyxy`` are the live reference transforms, ``build_assigner`` /
    ``build_sampler`` call the given factories (mmdet's assigner and sampler classes are absent from the tree)."""
    live = load_inference()
    name_s, name_o = f'{_PKG}.roi_heads.rotate_standard_roi_head', f'{_PKG}.roi_heads.oriented_standard_roi_head'
    T, _, _ = ref_rpn.load()

    class _Base(nn.Module):
        def __init__(self, init_cfg=None):
            super().__init__()
            self.init_cfg = init_cfg

    def build_head(cfg):
        cfg = dict(cfg)
        assert cfg.pop('type') == 'RotatedShared2FCBBoxHead'
        return live['shared2fc'](**cfg)

    def build_roi_extractor(cfg):
        cfg = dict(cfg)
        assert cfg.pop('type') == 'RotatedSingleRoIExtractor'
        return live['extractor'](**cfg)

    shims = {
        'mmcv': _mod('mmcv'), 'mmcv.runner': _mod('mmcv.runner', BaseModule=_Base),
        'mmdet': _mod('mmdet'), 'mmdet.core': _mod('mmdet.core', bbox2roi=None),
        'mmrotate': sys.modules.get('mmrotate') or _mod('mmrotate'),
        'mmrotate.core': _mod('mmrotate.core', build_assigner=assigner_factory, build_sampler=sampler_factory,
                              obb2xyxy=T.obb2xyxy, rbbox2result=None, rbbox2roi=T.rbbox2roi),
        f'{_PKG}.builder': _mod(f'{_PKG}.builder', ROTATED_HEADS=_Registry(), build_head=build_head,
                                build_roi_extractor=build_roi_extractor, build_shared_head=None),
    }
    saved = {k: sys.modules.get(k) for k in shims}
    sys.modules.update(shims)
    try:
        mods = []
        for name, parts in ((name_s, _ROI_HEAD_FILES[0]), (name_o, _ROI_HEAD_FILES[1])):
            spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, *parts))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[name] = mod
            spec.loader.exec_module(mod)
            mods.append(mod)
    finally:
        for k, v in saved.items():
            if k.startswith(_PKG) or k == 'mmrotate':
                continue
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mods[1].OrientedStandardRoIHead


_ASSIGNER_FILE = ('mmrotate', 'core', 'bbox', 'assigners', 'max_convex_iou_assigner.py')


def load_assign_rule():
    """-> the reference tree's ``MaxConvexIoUAssigner`` class, whose ``assign_wrt_overlaps``
    (max_convex_iou_assigner.py:124-207) is the MaxIoU assignment rule -- mmrotate's own copy of
    ``mmdet MaxIoUAssigner.assign_wrt_overlaps`` (negatives, positives, every box that ties a gt's best IoU), the one
    piece of the mmdet assigner that exists as CODE under /root/reference.  ``AssignResult`` is a plain container here."""
    name = f'{_PKG}_assign.max_convex_iou_assigner'
    if name in sys.modules:
        return sys.modules[name].MaxConvexIoUAssigner

    class AssignResult:
        def __init__(self, num_gts, gt_inds, max_overlaps, labels=None):
            self.num_gts, self.gt_inds, self.max_overlaps, self.labels = num_gts, gt_inds, max_overlaps, labels

    pkg = f'{_PKG}_assign'
    shims = {
        'mmcv': _mod('mmcv'), 'mmcv.ops': _mod('mmcv.ops', convex_iou=None),
        'mmdet': _mod('mmdet'), 'mmdet.core': _mod('mmdet.core'), 'mmdet.core.bbox': _mod('mmdet.core.bbox'),
        'mmdet.core.bbox.assigners': _mod('mmdet.core.bbox.assigners'),
        'mmdet.core.bbox.assigners.assign_result': _mod('mmdet.core.bbox.assigners.assign_result',
                                                        AssignResult=AssignResult),
        'mmdet.core.bbox.assigners.base_assigner': _mod('mmdet.core.bbox.assigners.base_assigner',
                                                        BaseAssigner=type('BaseAssigner', (), {})),
        pkg: _mod(pkg, __path__=[]),
        f'{_PKG}.builder': _mod(f'{_PKG}.builder', ROTATED_BBOX_ASSIGNERS=_Registry()),
    }
    # the file imports `..builder` relative to its package: give it a parent package of its own
    shims[f'{pkg}.builder'] = shims.pop(f'{_PKG}.builder')
    sub = f'{pkg}.assigners'
    shims[sub] = _mod(sub, __path__=[])
    name = f'{sub}.max_convex_iou_assigner'
    saved = {k: sys.modules.get(k) for k in shims}
    sys.modules.update(shims)
    try:
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, *_ASSIGNER_FILE))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if k.startswith(pkg):
                continue
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    sys.modules[f'{_PKG}_assign.max_convex_iou_assigner'] = mod
    return mod.MaxConvexIoUAssigner


_SAMPLER_FILE = ('mmrotate', 'core', 'bbox', 'samplers', 'rotate_random_sampler.py')


def load_sampler():
    """-> the reference's own ``RRandomSampler`` (rotate_random_sampler.py): its ``sample`` -- add_gt_as_proposals, the
    expected-positive / negative counts, neg_pos_ub -- and ``_sample_pos`` / ``_sample_neg`` are code of the reference
    tree; the mmdet pieces it leans on are stand-ins ([memory] mmdet 2.x): ``BaseSampler.__init__`` bookkeeping,
    ``SamplingResult`` as a container, ``AssignResult.add_gt_`` supplied by the caller's assign result."""
    name = f'{_PKG}_samplers.samplers.rotate_random_sampler'
    if name in sys.modules:
        return sys.modules[name].RRandomSampler

    class BaseSampler:
        def __init__(self, num, pos_fraction, neg_pos_ub=-1, add_gt_as_proposals=True, **kwargs):
            self.num, self.pos_fraction, self.neg_pos_ub = num, pos_fraction, neg_pos_ub
            self.add_gt_as_proposals = add_gt_as_proposals
            self.pos_sampler = self.neg_sampler = self

    class SamplingResult:
        def __init__(self, pos_inds, neg_inds, bboxes, gt_bboxes, assign_result, gt_flags):
            self.pos_inds, self.neg_inds, self.bboxes, self.gt_flags = pos_inds, neg_inds, bboxes, gt_flags

    pkg = f'{_PKG}_samplers'
    shims = {
        'mmdet': _mod('mmdet'), 'mmdet.core': _mod('mmdet.core'),
        'mmdet.core.bbox': _mod('mmdet.core.bbox', demodata=_mod('demodata', ensure_rng=lambda rng: rng)),
        'mmdet.core.bbox.samplers': _mod('mmdet.core.bbox.samplers'),
        'mmdet.core.bbox.samplers.base_sampler': _mod('mmdet.core.bbox.samplers.base_sampler', BaseSampler=BaseSampler),
        'mmdet.core.bbox.samplers.sampling_result': _mod('mmdet.core.bbox.samplers.sampling_result',
                                                         SamplingResult=SamplingResult),
        pkg: _mod(pkg, __path__=[]), f'{pkg}.builder': _mod(f'{pkg}.builder', ROTATED_BBOX_SAMPLERS=_Registry()),
        f'{pkg}.samplers': _mod(f'{pkg}.samplers', __path__=[]),
    }
    saved = {k: sys.modules.get(k) for k in shims}
    sys.modules.update(shims)
    keep = ('mmdet.core.bbox',)  # `from mmdet.core.bbox import demodata` runs inside __init__
    try:
        spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, *_SAMPLER_FILE))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if k.startswith(pkg) or k in keep or k in ('mmdet', 'mmdet.core'):
                continue
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod.RRandomSampler


_NMS_FILE = ('mmrotate', 'core', 'post_processing', 'bbox_nms_rotated.py')


def load_test_path(assigner_factory=None, sampler_factory=None):
    """-> dict(roi_head=<the reference's OrientedStandardRoIHead class, TEST-TIME path live>, multiclass_nms_rotated=<the
    reference's own function>, rbbox2result=<the reference's own>).  ``bbox_nms_rotated.py`` is imported unmodified with
    ``mmcv.ops.nms_rotated`` = the reference's python wrapper over the reference's CPU C++ (oracle/_ref); the module globals
    the import shims had stubbed -- ``rotated_bbox_head.multiclass_nms_rotated``, ``rotate_standard_roi_head.rbbox2result``
    -- are pointed at the live functions, so ``RotatedBBoxHead.get_bboxes`` (rotated_bbox_head.py:358-430),
    ``simple_test_bboxes`` (oriented_standard_roi_head.py:126-188) and ``simple_test`` (rotate_standard_roi_head.py:235-262)
    run as the reference wrote them."""
    live = load_inference()
    Head = load_roi_head(assigner_factory or (lambda cfg: None), sampler_factory or (lambda cfg, context=None: None))
    T, _, _ = ref_rpn.load()
    name = f'{_PKG}_post.bbox_nms_rotated'
    if name not in sys.modules:
        shims = {'mmcv': _mod('mmcv'), 'mmcv.ops': _mod('mmcv.ops', nms_rotated=live['ops']['nms'].nms_rotated)}
        saved = {k: sys.modules.get(k) for k in shims}
        sys.modules.update(shims)
        try:
            spec = importlib.util.spec_from_file_location(name, os.path.join(REF_ROOT, *_NMS_FILE))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[name] = mod
            spec.loader.exec_module(mod)
        finally:
            for k, v in saved.items():
                if v is None:
                    sys.modules.pop(k, None)
                else:
                    sys.modules[k] = v
    nms_mod = sys.modules[name]
    sys.modules[f'{_PKG}.roi_heads.bbox_heads.rotated_bbox_head'].multiclass_nms_rotated = nms_mod.multiclass_nms_rotated
    sys.modules[f'{_PKG}.roi_heads.rotate_standard_roi_head'].rbbox2result = T.rbbox2result
    return dict(roi_head=Head, multiclass_nms_rotated=nms_mod.multiclass_nms_rotated, rbbox2result=T.rbbox2result)
