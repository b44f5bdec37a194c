"""Import the REFERENCE backbone module (``mmrotate/models/backbones/convnext_moe.py``)
in this container, unmodified, from where it lies under ``/root/reference``.

TEST INFRASTRUCTURE ONLY.  Used by ``tests/golden/make_golden_moe.py`` to generate the
committed fixtures and by CPU tests (when ``/root/reference`` exists) to pin
``oracle/moe_oracle.py``.  It cannot travel to the GPU box -- nothing in the ``-m gpu``
tests, ``smoke()`` or ``bench.py`` may import it.

The reference file imports ``timm``, ``mmengine``, ``mmcv.cnn``, ``mmcv.runner`` and its
parent package's ``builder`` (-> ``mmdet``), none of which exist here.
They are replaced by the minimal stand-ins below, registered in ``sys.modules`` only for
the duration of the import:

* ``timm.models.layers.DropPath``      per-sample Bernoulli keep / keep_prob (timm semantics)
* ``timm.models.layers.trunc_normal_`` -> ``torch.nn.init.trunc_normal_``
* ``mmengine.model.{ModuleList,Sequential,BaseModule}`` -> torch containers
* ``mmcv.cnn.build_activation_layer``  -> ``nn.GELU()`` (act_cfg is GELU in every config)
* ``mmcv.runner.BaseModule``           -> ``nn.Module`` that stores ``init_cfg``
* ``<pkg>.builder.ROTATED_BACKBONES.register_module()`` -> identity decorator
"""
import importlib.util
import os
import sys
import types

import torch
import torch.nn as nn

REF_ROOT = os.environ.get('SM3DET_REFERENCE', '/root/reference')
REF_FILE = os.path.join(REF_ROOT, 'mmrotate', 'models', 'backbones', 'convnext_moe.py')
_PKG = '_sm3det_ref_pkg'


class _DropPath(nn.Module):
    def __init__(self, drop_prob=0.0, scale_by_keep=True):
        super().__init__()
        self.drop_prob = drop_prob
        self.scale_by_keep = scale_by_keep

    def forward(self, x):
        if self.drop_prob == 0.0 or not self.training:
            return x
        keep = 1 - self.drop_prob
        shape = (x.shape[0],) + (1,) * (x.ndim - 1)
        mask = x.new_empty(shape).bernoulli_(keep)
        if keep > 0.0 and self.scale_by_keep:
            mask.div_(keep)
        return x * mask


class _BaseModule(nn.Module):
    def __init__(self, init_cfg=None):
        super().__init__()
        self.init_cfg = init_cfg


class _Registry:
    def register_module(self, *a, **k):
        return lambda cls: cls


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    return m


REF_PYC = os.path.join(os.path.dirname(os.path.abspath(__file__)), '_ref', 'pyc', 'mmrotate.models.backbones.convnext_moe.pyc')


def available():
    """the reference source (build container) or its bytecode compiled by oracle/build_ref.py (travels to the GPU box)"""
    return os.path.exists(REF_FILE) or os.path.exists(REF_PYC)


def from_source():
    return os.path.exists(REF_FILE)


def load_reference_module():
    """Returns the imported reference module object (classes ConvNeXt_moe_MultiInput, MoE_layer...)."""
    if not available():
        raise FileNotFoundError(REF_FILE)
    full = f'{_PKG}.backbones.convnext_moe'
    if full in sys.modules:
        return sys.modules[full]

    def _gelu(cfg):
        assert cfg.get('type', 'GELU') == 'GELU'
        return nn.GELU()

    shims = {
        'timm': _mod('timm'),
        'timm.models': _mod('timm.models'),
        'timm.models.layers': _mod('timm.models.layers', DropPath=_DropPath,
                                   trunc_normal_=nn.init.trunc_normal_),
        'mmengine': _mod('mmengine'),
        'mmengine.model': _mod('mmengine.model', ModuleList=nn.ModuleList,
                               Sequential=nn.Sequential, BaseModule=_BaseModule),
        'mmengine.logging': _mod('mmengine.logging', MMLogger=type(
            'MMLogger', (), {'get_current_instance': staticmethod(lambda: None)})),
        'mmengine.runner': _mod('mmengine.runner'),
        'mmengine.runner.checkpoint': _mod('mmengine.runner.checkpoint',
                                           CheckpointLoader=object),
        'mmcv': _mod('mmcv'),
        'mmcv.cnn': _mod('mmcv.cnn', build_activation_layer=_gelu,
                         constant_init=lambda *a, **k: None,
                         trunc_normal_init=lambda *a, **k: None),
        'mmcv.runner': _mod('mmcv.runner', BaseModule=_BaseModule),
        _PKG: _mod(_PKG, __path__=[]),
        f'{_PKG}.builder': _mod(f'{_PKG}.builder', ROTATED_BACKBONES=_Registry()),
        f'{_PKG}.backbones': _mod(f'{_PKG}.backbones', __path__=[]),
    }
    saved = {k: sys.modules.get(k) for k in shims}
    sys.modules.update(shims)
    try:
        if from_source():
            spec = importlib.util.spec_from_file_location(full, REF_FILE)
        else:
            from importlib.machinery import SourcelessFileLoader
            loader = SourcelessFileLoader(full, REF_PYC)
            spec = importlib.util.spec_from_loader(full, loader)
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


def build_reference_backbone(**kwargs):
    """ConvNeXt_moe_MultiInput(**kwargs) with init_cfg=None."""
    mod = load_reference_module()
    kwargs.setdefault('init_cfg', None)
    return mod.ConvNeXt_moe_MultiInput(**kwargs)


if __name__ == '__main__':
    torch.manual_seed(0)
    net = build_reference_backbone(arch='tiny', MoE_Block_inds=[[], [0, 2], [0, 2, 4, 6, 8], [0, 2]],
                                   num_experts=8, top_k=2, drop_path_rate=0.1)
    print('params (M):', sum(p.numel() for p in net.parameters()) / 1e6)
    net.eval()
    outs, gl = net(torch.randn(1, 3, 128, 128), ['single'])
    print([tuple(o.shape) for o in outs], float(gl))
