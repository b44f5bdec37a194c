"""Minimal registry with the reference's plug-in surface for the backbone
(``@ROTATED_BACKBONES.register_module()`` + ``build(cfg)`` by ``type`` string;
mmrotate/models/builder.py:4-12, mmcv/mmcv/utils/registry.py:75).  When a real ``mmdet``/``mmcv`` registry is
importable, ``register_into(registry)`` adds the same classes to it so ``local_configs/main_SM3Det.py`` resolves
``type='ConvNeXt_moe_MultiInput'`` to the MI355X implementation."""
import inspect


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def __contains__(self, key):
        return key in self._module_dict

    def __len__(self):
        return len(self._module_dict)

    def _register(self, cls, name=None, force=False):
        if not inspect.isclass(cls):
            raise TypeError(f'module must be a class, but got {type(cls)}')
        name = name or cls.__name__
        if not force and name in self._module_dict:
            raise KeyError(f'{name} is already registered in {self._name}')
        self._module_dict[name] = cls

    def register_module(self, name=None, force=False, module=None):
        if module is not None:
            self._register(module, name, force)
            return module

        def _deco(cls):
            self._register(cls, name, force)
            return cls
        return _deco

    def build(self, cfg, default_args=None):
        if not isinstance(cfg, dict) or 'type' not in cfg:
            raise KeyError('`cfg` must be a dict containing the key "type"')
        args = dict(cfg)
        if default_args:
            for k, v in default_args.items():
                args.setdefault(k, v)
        t = args.pop('type')
        cls = self.get(t) if isinstance(t, str) else t
        if cls is None:
            raise KeyError(f'{t} is not in the {self._name} registry')
        return cls(**args)


ROTATED_BACKBONES = Registry('models')  # alias of mmdet's MODELS in the reference
MODELS = ROTATED_BACKBONES
ROTATED_NECKS = MODELS  # mmrotate/models/builder.py:4-12: every ROTATED_* name is the same mmdet MODELS registry


def register_into(registry, force=True):
    """Copy our classes into an external (mmcv/mmdet) registry object."""
    for name, cls in ROTATED_BACKBONES.module_dict.items():
        registry.register_module(name=name, force=force, module=cls)
