"""Import the REFERENCE's own config loader -- ``mmcv/mmcv/utils/config.py`` (``Config.fromfile``: python exec, ``_base_``
merge, ``_delete_``) -- unmodified from /root/reference, to pin sm3det_amd/config.py.  TEST INFRASTRUCTURE ONLY.

Stand-ins during the import: ``addict.Dict`` (absent from this image; [memory] the container semantics ConfigDict builds
on: nested dicts become Dicts, attribute access) and ``yapf`` (only used by ``pretty_text``, never called here); the
module's own ``.misc`` / ``.path`` siblings are the reference's real files."""
import importlib.util
import os
import sys
import types

from oracle.ref_moe import REF_ROOT

_PKG = '_sm3det_ref_mmcv_utils'


def available():
    return os.path.exists(os.path.join(REF_ROOT, 'mmcv', 'mmcv', 'utils', 'config.py'))


class _Dict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__()
        for arg in args:
            if not arg:
                continue
            if isinstance(arg, dict):
                for k, v in arg.items():
                    self[k] = self._hook(v)
            else:
                for k, v in iter(arg):
                    self[k] = self._hook(v)
        for k, v in kwargs.items():
            self[k] = self._hook(v)

    @classmethod
    def _hook(cls, item):
        if isinstance(item, dict):
            return cls(item)
        if isinstance(item, (list, tuple)):
            return type(item)(cls._hook(e) for e in item)
        return item

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, _Dict) else v) for k, v in self.items()}


def load():
    """-> the reference's `mmcv.utils.config` module (Config, ConfigDict)"""
    name = f'{_PKG}.config'
    if name in sys.modules:
        return sys.modules[name]
    if not available():
        raise FileNotFoundError(os.path.join(REF_ROOT, 'mmcv', 'mmcv', 'utils', 'config.py'))

    def _m(n, **kw):
        m = types.ModuleType(n)
        m.__dict__.update(kw)
        return m
    shims = {'addict': _m('addict', Dict=_Dict), 'yapf': _m('yapf'), 'yapf.yapflib': _m('yapf.yapflib'),
             'yapf.yapflib.yapf_api': _m('yapf.yapflib.yapf_api', FormatCode=lambda *a, **k: ('', False)),
             _PKG: _m(_PKG, __path__=[])}
    saved = {k: sys.modules.get(k) for k in shims}
    sys.modules.update(shims)
    try:
        mods = {}
        for sub in ('misc', 'path', 'config'):
            full = f'{_PKG}.{sub}'
            spec = importlib.util.spec_from_file_location(full, os.path.join(REF_ROOT, 'mmcv', 'mmcv', 'utils', sub + '.py'))
            mod = importlib.util.module_from_spec(spec)
            sys.modules[full] = mod
            spec.loader.exec_module(mod)
            mods[sub] = mod
    finally:
        for k, v in saved.items():
            if k.startswith(_PKG):
                continue
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v
    return mods['config']
