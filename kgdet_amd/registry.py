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

    def __init__(self, name):
        self._name = name
        self._module_dict = dict()

    def __repr__(self):
        return '{}(name={}, items={})'.format(self.__class__.__name__, self._name,
                                               list(self._module_dict.keys()))

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key, None)

    def _register_module(self, module_class, name=None):
        if not inspect.isclass(module_class):
            raise TypeError('module must be a class, but got {}'.format(type(module_class)))
        module_name = name or module_class.__name__
        if module_name in self._module_dict:
            raise KeyError('{} is already registered in {}'.format(module_name, self.name))
        self._module_dict[module_name] = module_class

    def register_module(self, cls):
        self._register_module(cls)
        return cls

    def register_alias(self, name, cls):
        """Extra name for an already registered class (e.g. ``KGDetHead``)."""
        self._register_module(cls, name)


def build_from_cfg(cfg, registry, default_args=None):
    assert isinstance(cfg, dict) and 'type' in cfg
    assert isinstance(default_args, dict) or default_args is None
    args = dict(cfg)
    obj_type = args.pop('type')
    if isinstance(obj_type, str):
        found = registry.get(obj_type)
        if found is None:
            raise KeyError('{} is not in the {} registry'.format(obj_type, registry.name))
        obj_type = found
    elif not inspect.isclass(obj_type):
        raise TypeError('type must be a str or valid type, but got {}'.format(type(obj_type)))
    if default_args is not None:
        for name, value in default_args.items():
            args.setdefault(name, value)
    return obj_type(**args)


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
