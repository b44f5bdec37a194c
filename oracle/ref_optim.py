"""Import the REFERENCE's own ``DefaultOptimizerConstructor`` (mmcv/mmcv/runner/optimizer/default_constructor.py)
unmodified from /root/reference, to pin sm3det_amd/optim.py's `reference_param_options` / `param_groups_from_cfg`.
TEST INFRASTRUCTURE ONLY.  Stand-ins during the import: ``mmcv.utils`` names it uses (``_BatchNorm`` / ``_InstanceNorm``
= torch's classes, ``is_list_of``, ``build_from_cfg`` = construct ``torch.optim.<type>``), ``check_ops_exist`` -> False
(no DCN layer in the SM3Det models), the two registries of ``.builder``."""
import importlib.util
import os
import sys
import types

import torch

from oracle.ref_moe import REF_ROOT

_PKG = '_sm3det_ref_mmcv_optim'
_FILE = os.path.join(REF_ROOT, 'mmcv', 'mmcv', 'runner', 'optimizer', 'default_constructor.py')


def available():
    return os.path.exists(_FILE)


class _Registry:
    def register_module(self, *a, **k):
        return lambda cls: cls


def _build_from_cfg(cfg, registry):
    cfg = dict(cfg)
    return getattr(torch.optim, cfg.pop('type'))(**cfg)


def load():
    name = f'{_PKG}.default_constructor'
    if name in sys.modules:
        return sys.modules[name].DefaultOptimizerConstructor
    if not available():
        raise FileNotFoundError(_FILE)

    def _m(n, **kw):
        m = types.ModuleType(n)
        m.__dict__.update(kw)
        return m
    shims = {
        'mmcv': _m('mmcv'),
        'mmcv.utils': _m('mmcv.utils', _BatchNorm=torch.nn.modules.batchnorm._BatchNorm,
                         _InstanceNorm=torch.nn.modules.instancenorm._InstanceNorm, build_from_cfg=_build_from_cfg,
                         is_list_of=lambda seq, t: isinstance(seq, list) and all(isinstance(x, t) for x in seq)),
        'mmcv.utils.ext_loader': _m('mmcv.utils.ext_loader', check_ops_exist=lambda: False),
        _PKG: _m(_PKG, __path__=[]),
        f'{_PKG}.builder': _m(f'{_PKG}.builder', OPTIMIZER_BUILDERS=_Registry(), OPTIMIZERS=_Registry()),
    }
    saved = {k: sys.modules.get(k) for k in shims}
    sys.modules.update(shims)
    try:
        spec = importlib.util.spec_from_file_location(name, _FILE)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if k.startswith(_PKG):
                continue
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mod.DefaultOptimizerConstructor
