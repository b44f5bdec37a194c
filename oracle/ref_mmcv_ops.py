"""Load the REFERENCE's own python operator wrappers (``mmcv/mmcv/ops/{box_iou_rotated,nms,roi_align_rotated,
deform_conv}.py``) unmodified and bind them to ``sm3det_amd.mmcv_ext`` installed as ``mmcv._ext`` -- the drop-in test of
SURVEY.md 8(b).1: the wrappers obtain their native functions with ``ext_loader.load_ext('_ext', [...])``.

TEST INFRASTRUCTURE ONLY.  Sources: the files where they lie under ``/root/reference`` when that tree exists (build
container), otherwise the bytecode ``oracle/build_ref.py`` compiled from them into ``oracle/_ref/pyc`` (git-ignored,
travels to the GPU box).  Nothing of the reference is copied into the repo.

The wrappers import ``mmcv.utils`` (whose ``__init__`` needs addict/yapf/cv2, absent here) and ``mmcv.cnn.CONV_LAYERS``;
a stand-in ``mmcv`` package is assembled in ``sys.modules`` from the reference's REAL ``utils/ext_loader.py`` and
``utils/misc.py`` (``load_ext``, ``deprecated_api_warning``) plus a two-line ``CONV_LAYERS`` / ``print_log``.
"""
import importlib.machinery
import importlib.util
import os
import sys
import types

from oracle import build_ref

_LOADED = None


def available():
    root = os.path.join(build_ref.REF_ROOT, 'mmcv', 'mmcv')
    return os.path.isdir(root) or os.path.exists(os.path.join(build_ref.PYC, 'mmcv.ops.nms.pyc'))


def _load(mod_name, rel):
    src = os.path.join(build_ref.REF_ROOT, 'mmcv', 'mmcv', rel)
    if os.path.exists(src):
        loader = importlib.machinery.SourceFileLoader(mod_name, src)
    else:
        loader = importlib.machinery.SourcelessFileLoader(mod_name, os.path.join(build_ref.PYC, mod_name + '.pyc'))
    spec = importlib.util.spec_from_loader(mod_name, loader)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[mod_name] = mod
    loader.exec_module(mod)
    return mod


class _Registry:
    def register_module(self, *a, **k):
        return lambda cls: cls


def load(ext=None):
    """-> dict of the reference wrapper modules {'box_iou_rotated', 'nms', 'roi_align_rotated', 'deform_conv'}, bound to
    `ext` (default: sm3det_amd.mmcv_ext) as `mmcv._ext`.  Registers stand-in `mmcv*` modules in sys.modules."""
    global _LOADED
    if ext is None:
        from sm3det_amd import mmcv_ext
        ext = mmcv_ext.install_as_mmcv_ext()
    else:
        sys.modules['mmcv._ext'] = ext
    if _LOADED is not None and _LOADED[0] is ext:
        return _LOADED[1]
    for k in [k for k in sys.modules if k == 'mmcv' or (k.startswith('mmcv.') and k != 'mmcv._ext')]:
        del sys.modules[k]
    pkg = types.ModuleType('mmcv')
    pkg.__path__ = []
    sys.modules['mmcv'] = pkg
    pkg._ext = ext
    utils = types.ModuleType('mmcv.utils')
    utils.__path__ = []
    sys.modules['mmcv.utils'] = utils
    utils.ext_loader = _load('mmcv.utils.ext_loader', 'utils/ext_loader.py')
    misc = _load('mmcv.utils.misc', 'utils/misc.py')
    utils.deprecated_api_warning = misc.deprecated_api_warning
    utils.print_log = lambda *a, **k: None
    pkg.utils = utils
    cnn = types.ModuleType('mmcv.cnn')
    cnn.CONV_LAYERS = _Registry()
    sys.modules['mmcv.cnn'] = cnn
    ops = types.ModuleType('mmcv.ops')
    ops.__path__ = []
    sys.modules['mmcv.ops'] = ops
    out = {}
    for name in ('box_iou_rotated', 'nms', 'roi_align_rotated', 'deform_conv'):
        out[name] = _load('mmcv.ops.' + name, f'ops/{name}.py')
        setattr(ops, name, out[name])
    _LOADED = (ext, out)
    return out
