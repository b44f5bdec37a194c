"""Import the REFERENCE detector class -- ``mmrotate/models/detectors/trisource_H1stage_R2stage_detector.py``
(``TriSourceDetector``) -- unmodified from /root/reference, so that its own ``__init__`` / ``extract_feat`` /
``split_batch`` / ``gather_dict_values`` / ``forward_train`` code pins the composition of ``sm3det_amd/detector.py``.

TEST INFRASTRUCTURE ONLY (CPU tests in the build container; the GPU box has no /root/reference).

Stand-ins registered in ``sys.modules`` for the duration of the import (mmdet is not importable here, SURVEY.md Appendix
A): ``mmdet.core.bbox2result`` (test-time only, never called), ``<pkg>.builder.{ROTATED_DETECTORS, build_backbone,
build_head, build_neck}`` -- the three builders go through a registry the CALLER supplies (the tests register small
recording stubs there, and the same registry builds the product detector's children) -- and
``<pkg>.detectors.base.RotatedBaseDetector`` = an ``nn.Module`` with mmdet ``BaseDetector``'s ``with_neck`` property."""
import importlib.util
import os
import sys

import torch.nn as nn

from oracle.ref_moe import REF_ROOT, _Registry, _mod

_PKG = '_sm3det_ref_pkg_det'
REF_FILE = os.path.join(REF_ROOT, 'mmrotate', 'models', 'detectors', 'trisource_H1stage_R2stage_detector.py')


def available():
    return os.path.exists(REF_FILE)


class _Base(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg
        self.fp16_enabled = False

    @property
    def with_neck(self):  # mmdet BaseDetector.with_neck
        return hasattr(self, 'neck') and self.neck is not None


def load(registry):
    """-> the reference module; `registry.build(cfg)` constructs every child (backbone, neck, heads)"""
    if not available():
        raise FileNotFoundError(REF_FILE)
    build = lambda cfg: registry.build(dict(cfg))  # noqa: E731

    def _never(*a, **k):
        raise AssertionError('test-time only')

    name = f'{_PKG}.detectors.trisource'
    shims = {
        'mmdet': _mod('mmdet'), 'mmdet.core': _mod('mmdet.core', bbox2result=_never),
        _PKG: _mod(_PKG, __path__=[]),
        f'{_PKG}.builder': _mod(f'{_PKG}.builder', ROTATED_DETECTORS=_Registry(), build_backbone=build, build_head=build,
                                build_neck=build),
        f'{_PKG}.detectors': _mod(f'{_PKG}.detectors', __path__=[]),
        f'{_PKG}.detectors.base': _mod(f'{_PKG}.detectors.base', RotatedBaseDetector=_Base),
    }
    saved = {k: sys.modules.get(k) for k in shims}
    sys.modules.update(shims)
    try:
        spec = importlib.util.spec_from_file_location(name, REF_FILE)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
        sys.modules.pop(name, None)
    return mod
