"""Import the REFERENCE box transforms and MidpointOffset coder (``mmrotate/core/bbox/transforms.py``,
``mmrotate/core/bbox/coder/delta_midpointoffset_rbbox_coder.py``) unmodified from /root/reference.  TEST
INFRASTRUCTURE ONLY.  Stand-ins during the import: ``cv2`` (only used by the numpy helpers, never called here),
``mmcv.jit`` -> identity decorator, ``mmdet...BaseBBoxCoder`` -> object, ``<pkg>.builder.ROTATED_BBOX_CODERS`` ->
identity registry."""
import importlib.util
import os
import sys

from oracle.ref_moe import REF_ROOT, _Registry, _mod

_PKG = '_sm3det_ref_pkg_bbox'
T_FILE = os.path.join(REF_ROOT, 'mmrotate', 'core', 'bbox', 'transforms.py')
C_FILE = os.path.join(REF_ROOT, 'mmrotate', 'core', 'bbox', 'coder', 'delta_midpointoffset_rbbox_coder.py')
X_FILE = os.path.join(REF_ROOT, 'mmrotate', 'core', 'bbox', 'coder', 'delta_xywha_rbbox_coder.py')


def available():
    return os.path.exists(T_FILE) and os.path.exists(C_FILE)


def load():
    """-> (transforms module, MidpointOffset coder module, DeltaXYWHA coder module)"""
    if not available():
        raise FileNotFoundError(T_FILE)
    tn, cn = f'{_PKG}.transforms', f'{_PKG}.coder.delta_midpointoffset_rbbox_coder'
    xn = f'{_PKG}.coder.delta_xywha_rbbox_coder'
    if xn in sys.modules:
        return sys.modules[tn], sys.modules[cn], sys.modules[xn]
    jit = lambda *a, **k: (lambda f: f)  # noqa: E731
    shims = {
        'cv2': _mod('cv2'),
        'mmcv': _mod('mmcv', jit=jit),
        'mmdet': _mod('mmdet'), 'mmdet.core': _mod('mmdet.core'), 'mmdet.core.bbox': _mod('mmdet.core.bbox'),
        'mmdet.core.bbox.coder': _mod('mmdet.core.bbox.coder'),
        'mmdet.core.bbox.coder.base_bbox_coder': _mod('mmdet.core.bbox.coder.base_bbox_coder',
                                                      BaseBBoxCoder=type('BaseBBoxCoder', (), {
                                                          '__init__': lambda self, **kw: None})),
        _PKG: _mod(_PKG, __path__=[]),
        f'{_PKG}.builder': _mod(f'{_PKG}.builder', ROTATED_BBOX_CODERS=_Registry()),
        f'{_PKG}.coder': _mod(f'{_PKG}.coder', __path__=[]),
    }
    saved = {k: sys.modules.get(k) for k in shims}
    sys.modules.update(shims)
    try:
        mods = []
        for name, path in ((tn, T_FILE), (cn, C_FILE), (xn, X_FILE)):
            spec = importlib.util.spec_from_file_location(name, path)
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
