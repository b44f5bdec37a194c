"""Minimal loader for the reference's python-file configs (``mmcv.Config.fromfile`` semantics,
mmcv/mmcv/utils/config.py:181-330) so that ``local_configs/main_SM3Det.py`` and friends are consumed UNCHANGED:

* the file is executed as Python (the configs use list comprehensions and cross-variable arithmetic); public,
  non-module, non-callable top-level names become the config dict;
* ``_base_`` (a path or list of paths relative to the file) is loaded first, bases must not share top-level keys, the
  child is merged over them key by key, ``_delete_=True`` in a child dict replaces instead of merging;
* a ``_base_`` path that does not exist is retried against ``<tree>/configs/_base_/<tail after '_base_/'>``: several
  files of the reference carry the ``_base_`` prefix that is correct for the *other* of its two config directories
  (``local_configs/SM3Det_convnext_b.py:1-4``).

``build_detector_pieces(cfg.model)`` then constructs every sub-module this package implements through the registry by
its ``type`` string with the config dict passed through unchanged.
"""
import os
import types


class ConfigDict(dict):
    """dict with attribute access (the subset of addict.Dict the reference code relies on)"""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k) from None

    def __setattr__(self, k, v):
        self[k] = v


def _wrap(v):
    if isinstance(v, dict):
        return ConfigDict({k: _wrap(x) for k, x in v.items()})
    if isinstance(v, (list, tuple)):
        return type(v)(_wrap(x) for x in v)
    return v


def _merge(child, base):
    out = dict(base)
    for k, v in child.items():
        if isinstance(v, dict):
            v = dict(v)
            delete = v.pop('_delete_', False)
            if k in out and not delete:
                if not isinstance(out[k], dict):
                    raise TypeError(f'{k}={v} in child config cannot inherit from base because {k} is a dict in the '
                                    f'child config but is of type {type(out[k])} in base config. You may set '
                                    '`_delete_=True` to ignore the base config.')
                out[k] = _merge(v, out[k])
            else:
                out[k] = _merge(v, {})
        else:
            out[k] = v
    return out


def _resolve_base(path, here):
    cand = os.path.normpath(os.path.join(here, path))
    if os.path.exists(cand):
        return cand
    if '_base_/' in path:  # documented fallback: <tree>/configs/_base_/<tail>
        tail = path.split('_base_/', 1)[1]
        d = here
        for _ in range(4):
            alt = os.path.join(d, 'configs', '_base_', tail)
            if os.path.exists(alt):
                return alt
            d = os.path.dirname(d)
    raise FileNotFoundError(f'_base_ file {path!r} of a config in {here} not found')


def _file2dict(filename):
    filename = os.path.abspath(filename)
    ns = {'__file__': filename}
    with open(filename) as f:
        exec(compile(f.read(), filename, 'exec'), ns)  # noqa: S102 -- configs are Python by design (mmcv does the same)
    cfg = {k: v for k, v in ns.items()
           if not k.startswith('__') and not isinstance(v, (types.ModuleType, types.FunctionType, type))}
    base = cfg.pop('_base_', None)
    if base is None:
        return cfg
    merged = {}
    for b in ([base] if isinstance(base, str) else list(base)):
        c = _file2dict(_resolve_base(b, os.path.dirname(filename)))
        dup = merged.keys() & c.keys()
        if dup:
            raise KeyError(f'Duplicate key is not allowed among bases. Duplicate keys: {dup}')
        merged.update(c)
    return _merge(cfg, merged)


class Config(ConfigDict):
    @staticmethod
    def fromfile(filename):
        cfg = Config(_wrap(_file2dict(filename)))
        dict.__setitem__(cfg, 'filename', os.path.abspath(filename))
        return cfg


# sub-modules of `model` built by this package; everything else in the config (train/test cfgs, the mmdet-side loss
# and assigner dicts nested inside the heads) travels inside the dicts untouched
_PIECES = ('backbone', 'neck', 'sar_bbox_head', 'rgb_rpn_head', 'ifr_rpn_head')
_ROI_PIECES = ('bbox_roi_extractor', 'bbox_head')


def build_detector_pieces(model_cfg, skip_missing=True):
    """{name: module} for every piece of a ``TriSourceDetector`` model dict whose ``type`` is registered here.
    The dicts are passed to the constructors unchanged (``init_cfg`` of the backbone included: nothing is loaded
    until ``init_weights()`` is called, as in the reference)."""
    from . import fpn, roi_head, rpn_head  # noqa: F401  (registration side effects)
    try:
        from . import gfl_head  # noqa: F401
    except ImportError:
        pass
    from .registry import MODELS
    out = {}

    def build(name, cfg):
        if cfg is None:
            return
        if cfg['type'] not in MODELS:
            if skip_missing:
                return
            raise KeyError(f"{cfg['type']} is not in the {MODELS.name} registry")
        out[name] = MODELS.build(cfg)
    for name in _PIECES:
        cfg = model_cfg.get(name)
        if cfg is not None and name.endswith('_rpn_head'):
            # TriSourceDetector.__init__ (trisource_H1stage_R2stage_detector.py:64-68): the modality's rpn train / test cfg
            mod = name.split('_')[0]
            tr, te = model_cfg.get(f'{mod}_train_cfg'), model_cfg.get(f'{mod}_test_cfg')
            cfg = dict(cfg, train_cfg=(tr or {}).get('rpn'), test_cfg=(te or {}).get('rpn'))
        elif cfg is not None and name == 'sar_bbox_head':
            # (:100-104): the one-stage head takes the SAR train / test cfg whole
            cfg = dict(cfg, train_cfg=model_cfg.get('sar_train_cfg'), test_cfg=model_cfg.get('sar_test_cfg'))
        build(name, cfg)
    for head in ('rgb_roi_head', 'ifr_roi_head'):
        h = model_cfg.get(head)
        if h is not None:
            mod = head.split('_')[0]  # (:70-77): the whole RoI head with the modality's rcnn train / test cfg
            tr, te = model_cfg.get(f'{mod}_train_cfg'), model_cfg.get(f'{mod}_test_cfg')
            build(head, dict(h, train_cfg=(tr or {}).get('rcnn'), test_cfg=(te or {}).get('rcnn')))
            if head in out:
                # the RoI head is built ONCE; its sub-modules are exposed as ALIASES of it (one parameter set per reference
                # key: a checkpoint loaded into / an optimizer built over this dict sees each weight exactly once)
                for name in _ROI_PIECES:
                    sub = getattr(out[head], name, None)
                    if sub is not None:
                        out[f'{head}.{name}'] = sub
            else:  # the head's own type is not registered: build what is
                for name in _ROI_PIECES:
                    build(f'{head}.{name}', h.get(name))
    return out
