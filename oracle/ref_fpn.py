"""Import the REFERENCE neck (``mmrotate/models/necks/Multitask_FPN.py``) in this container, unmodified, from where
it lies under ``/root/reference``.  TEST INFRASTRUCTURE ONLY (fixture generation + CPU pinning tests; cannot travel to
the GPU box).

The file imports ``mmcv.runner``, ``mmcv.cnn`` and ``mmengine.model`` (not importable here) and its
parent package's ``builder``.  Stand-ins, registered in ``sys.modules`` only during the import:

* ``mmcv.cnn.ConvModule`` -> conv + bias with the ``.conv`` child (what the real ConvModule is when
  conv_cfg = norm_cfg = act_cfg = None, mmcv/mmcv/cnn/bricks/conv_module.py:70-160); other cfgs assert.
* ``mmcv.runner.{BaseModule, auto_fp16}`` / ``mmengine.model.BaseModule`` -> nn.Module storing init_cfg / identity
  decorator;  ``<pkg>.builder.ROTATED_NECKS.register_module()`` -> identity decorator.
"""
import importlib.util
import os
import sys

import torch.nn as nn

from oracle.ref_moe import REF_ROOT, _BaseModule, _Registry, _mod

REF_FILE = os.path.join(REF_ROOT, 'mmrotate', 'models', 'necks', 'Multitask_FPN.py')
_PKG = '_sm3det_ref_pkg_necks'


class _ConvModule(nn.Module):
    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, conv_cfg=None, norm_cfg=None,
                 act_cfg=None, inplace=False):
        super().__init__()
        assert conv_cfg is None and norm_cfg is None and act_cfg is None
        self.conv = nn.Conv2d(in_channels, out_channels, kernel_size, stride=stride, padding=padding)

    def forward(self, x):
        return self.conv(x)


def available():
    return os.path.exists(REF_FILE)


def load_reference_module():
    if not available():
        raise FileNotFoundError(REF_FILE)
    full = f'{_PKG}.necks.Multitask_FPN'
    if full in sys.modules:
        return sys.modules[full]

    def auto_fp16(*a, **k):
        return lambda f: f

    shims = {
        'mmengine': _mod('mmengine'),
        'mmengine.model': _mod('mmengine.model', BaseModule=_BaseModule),
        'mmcv': _mod('mmcv'),
        'mmcv.cnn': _mod('mmcv.cnn', ConvModule=_ConvModule, build_norm_layer=lambda *a, **k: None),
        'mmcv.runner': _mod('mmcv.runner', BaseModule=_BaseModule, auto_fp16=auto_fp16),
        _PKG: _mod(_PKG, __path__=[]),
        f'{_PKG}.builder': _mod(f'{_PKG}.builder', ROTATED_NECKS=_Registry()),
        f'{_PKG}.necks': _mod(f'{_PKG}.necks', __path__=[]),
    }
    saved = {k: sys.modules.get(k) for k in shims}
    sys.modules.update(shims)
    try:
        spec = importlib.util.spec_from_file_location(full, REF_FILE)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[full] = mod
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if k.startswith(_PKG):
                continue
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod
