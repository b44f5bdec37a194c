"""Import the REFERENCE dynamic-lr hook (``mmrotate/core/hook/dynamic_lr.py``) unmodified from /root/reference, on top of
the reference's OWN ``LrUpdaterHook`` (``mmcv/mmcv/runner/hooks/lr_updater.py``, also loaded unmodified: ``load(real_base=
True)``, the default since round 6 -- the warm-up behaviour of an iteration-based run lives in that base class).  TEST
INFRASTRUCTURE ONLY (pins ``sm3det_amd.optim.DynamicLrPolicy`` / ``dynamic_lr_after_train_iter`` on CPU).  Stand-ins during
the import: ``mmcv.is_list_of``, ``mmcv.runner.BaseRunner``, the ``Hook`` base class (no method of it is reached) and an
identity ``HOOKS`` registry.  ``load(real_base=False)`` keeps round 5's restated base (``_LrUpdaterHook`` below), which
implements the warm-up mmcv's docstring describes rather than what an IterBasedRunner run does."""
import importlib.util
import os
import sys

from oracle.ref_moe import REF_ROOT, _Registry, _mod

REF_FILE = os.path.join(REF_ROOT, 'mmrotate', 'core', 'hook', 'dynamic_lr.py')


class _LrUpdaterHook:
    """what the dynamic hook uses of mmcv's LrUpdaterHook: the constructor fields, ``get_warmup_lr`` for warmup='linear'
    (mmcv/mmcv/runner/hooks/lr_updater.py:75-92, restated: k = (1 - cur / warmup_iters) * (1 - warmup_ratio);
    lr = regular * (1 - k)) and a ``_set_lr`` that records what would be written into the optimizer's groups"""

    def __init__(self, by_epoch=True, warmup=None, warmup_iters=0, warmup_ratio=0.1, warmup_by_epoch=False):
        self.by_epoch, self.warmup, self.warmup_iters, self.warmup_ratio = by_epoch, warmup, warmup_iters, warmup_ratio
        self.base_lr, self.regular_lr = [], []
        self.last_set = None

    def get_warmup_lr(self, cur_iters):
        assert self.warmup == 'linear'
        k = (1 - cur_iters / self.warmup_iters) * (1 - self.warmup_ratio)
        return [_lr * (1 - k) for _lr in self.regular_lr]

    def _set_lr(self, runner, lr_groups):
        self.last_set = list(lr_groups)


def available():
    return os.path.exists(REF_FILE)


LR_FILE = os.path.join(REF_ROOT, 'mmcv', 'mmcv', 'runner', 'hooks', 'lr_updater.py')


def _real_lr_updater():
    """the reference's own mmcv LrUpdaterHook class, from its file"""
    name = '_sm3det_ref_mmcv_hooks.lr_updater'
    if name in sys.modules:
        return sys.modules[name].LrUpdaterHook
    pkg = _mod('_sm3det_ref_mmcv_hooks')
    pkg.__path__ = []
    hook = _mod('_sm3det_ref_mmcv_hooks.hook', HOOKS=_Registry(), Hook=type('Hook', (), {}))
    shims = {'_sm3det_ref_mmcv_hooks': pkg, '_sm3det_ref_mmcv_hooks.hook': hook,
             'mmcv': _mod('mmcv', is_list_of=lambda seq, t: isinstance(seq, list) and all(isinstance(x, t) for x in seq)),
             'mmcv.runner': _mod('mmcv.runner', BaseRunner=object)}
    shims['mmcv'].runner = shims['mmcv.runner']
    saved = {k: sys.modules.get(k) for k in shims}
    sys.modules.update(shims)
    try:
        spec = importlib.util.spec_from_file_location(name, LR_FILE)
        mod = importlib.util.module_from_spec(spec)
        mod.__package__ = '_sm3det_ref_mmcv_hooks'
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
    finally:
        for k, v in saved.items():
            if k.startswith('mmcv'):
                if v is None:
                    sys.modules.pop(k, None)
                else:
                    sys.modules[k] = v
    return mod.LrUpdaterHook


def load(real_base=True):
    if not available():
        raise FileNotFoundError(REF_FILE)
    name = '_sm3det_ref_dynamic_lr' + ('' if real_base else '_restated_base')
    if name in sys.modules:
        return sys.modules[name]
    hooks = _mod('mmcv.runner.hooks', LrUpdaterHook=_real_lr_updater() if real_base else _LrUpdaterHook)
    shims = {
        'mmcv': _mod('mmcv', is_list_of=lambda seq, t: isinstance(seq, list) and all(isinstance(x, t) for x in seq)),
        'mmcv.runner': _mod('mmcv.runner', hooks=hooks, BaseRunner=object),
        'mmcv.runner.hooks': hooks,
        'mmcv.runner.hooks.hook': _mod('mmcv.runner.hooks.hook', HOOKS=_Registry(), Hook=object),
    }
    shims['mmcv'].runner = shims['mmcv.runner']
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
    return mod
