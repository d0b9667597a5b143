"""Registry / config-dict factory -- the plug-in API the hot path sits behind.

Mirrors mmdet/utils/registry.py:28-74 (``Registry``, ``build_from_cfg``), mmdet/models/registry.py:3-9
(the seven model registries) and mmdet/models/builder.py:8-43 (``build_*``), plus the slice of
``mmcv.Config`` the reference relies on: Python-file configs evaluated into nested dicts with
attribute access (``cfg.uniform.assigner['type']``, ``cfg.get('nms_pre', -1)``).
"""
import inspect
import os
import runpy

from torch import nn


class Registry(object):
    """name -> class table behind ``@X.register_module`` and ``build_from_cfg`` (the behaviour of
    mmdet/utils/registry.py:6-46: duplicate names and non-classes are errors, ``get`` returns None for unknown names)."""

    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    name = property(lambda self: self._name)
    module_dict = property(lambda self: self._module_dict)

    def __repr__(self):
        return '%s(name=%s, items=%s)' % (type(self).__name__, self._name, list(self._module_dict))

    def get(self, key):
        return self._module_dict.get(key)

    def _register_module(self, module_class, name=None):
        if not inspect.isclass(module_class):
            raise TypeError('module must be a class, but got {}'.format(type(module_class)))
        key = module_class.__name__ if name is None else name
        if key in self._module_dict:
            raise KeyError('{} is already registered in {}'.format(key, self._name))
        self._module_dict[key] = module_class

    def register_module(self, cls):
        """class decorator"""
        self._register_module(cls)
        return cls

    def register_alias(self, name, cls):
        """Extra name for an already registered class (e.g. ``KGDetHead``)."""
        self._register_module(cls, name)


def build_from_cfg(cfg, registry, default_args=None):
    """Instantiate ``cfg['type']`` (a registered name or a class) with the remaining keys; ``default_args`` fill the
    keys the config leaves out (registry.py:49-74)."""
    if not (isinstance(cfg, dict) and 'type' in cfg):
        raise AssertionError('cfg must be a dict with a "type" key')
    if not (default_args is None or isinstance(default_args, dict)):
        raise AssertionError('default_args must be a dict or None')
    kwargs = {k: v for k, v in cfg.items() if k != 'type'}
    target = cfg['type']
    if isinstance(target, str):
        cls = registry.get(target)
        if cls is None:
            raise KeyError('{} is not in the {} registry'.format(target, registry.name))
    elif inspect.isclass(target):
        cls = target
    else:
        raise TypeError('type must be a str or valid type, but got {}'.format(type(target)))
    for key, value in (default_args or {}).items():
        kwargs.setdefault(key, value)
    return cls(**kwargs)


BACKBONES = Registry('backbone')
NECKS = Registry('neck')
ROI_EXTRACTORS = Registry('roi_extractor')
SHARED_HEADS = Registry('shared_head')
HEADS = Registry('head')
LOSSES = Registry('loss')
DETECTORS = Registry('detector')
DATASETS = Registry('dataset')


def build(cfg, registry, default_args=None):
    if isinstance(cfg, list):
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_backbone(cfg):
    return build(cfg, BACKBONES)


def build_neck(cfg):
    return build(cfg, NECKS)


def build_roi_extractor(cfg):
    return build(cfg, ROI_EXTRACTORS)


def build_shared_head(cfg):
    return build(cfg, SHARED_HEADS)


def build_head(cfg):
    return build(cfg, HEADS)


def build_loss(cfg):
    return build(cfg, LOSSES)


def build_detector(cfg, train_cfg=None, test_cfg=None):
    return build(cfg, DETECTORS, dict(train_cfg=train_cfg, test_cfg=test_cfg))


# ------------------------------------------------------------------------------------------------
class ConfigDict(dict):
    """dict with attribute access, nested (what the head reads ``train_cfg`` / ``test_cfg`` through)."""

    def __init__(self, *args, **kwargs):
        super(ConfigDict, self).__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            return ConfigDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(ConfigDict._wrap(e) for e in v)
        return v

    def __setitem__(self, k, v):
        super(ConfigDict, self).__setitem__(k, ConfigDict._wrap(v))

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError("'ConfigDict' object has no attribute '{}'".format(name))

    def __setattr__(self, name, value):
        self[name] = value

    def copy(self):
        return ConfigDict(self)


class Config(object):
    """``Config.fromfile('configs/kgdet_moment_r50_fpn_1x-demo.py')`` -> attribute-style config."""

    def __init__(self, cfg_dict=None, filename=None):
        super(Config, self).__setattr__('_cfg_dict', ConfigDict(cfg_dict or {}))
        super(Config, self).__setattr__('_filename', filename)

    @staticmethod
    def fromfile(filename):
        filename = os.path.abspath(os.path.expanduser(filename))
        if not os.path.isfile(filename):
            raise FileNotFoundError('file "{}" does not exist'.format(filename))
        if not filename.endswith('.py'):
            raise IOError('Only py type config files are supported')
        ns = runpy.run_path(filename)
        cfg = {k: v for k, v in ns.items()
               if not k.startswith('__') and not inspect.ismodule(v) and not inspect.isfunction(v)}
        return Config(cfg, filename=filename)

    @property
    def filename(self):
        return self._filename

    def __getattr__(self, name):
        return getattr(self._cfg_dict, name)

    def __getitem__(self, name):
        return self._cfg_dict[name]

    def __setattr__(self, name, value):
        self._cfg_dict[name] = value

    def __contains__(self, name):
        return name in self._cfg_dict

    def get(self, key, default=None):
        return self._cfg_dict.get(key, default)

    def __repr__(self):
        return 'Config (path: {}): {}'.format(self._filename, dict.__repr__(self._cfg_dict))
