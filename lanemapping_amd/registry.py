"""Registry / builder mirror of the reference's plugin interface (the drop-in boundary).

Same API and error behaviour as baseline/utils/registry.py:12-82 and baseline/models/registry.py:5-36:
five model registries keyed by class ``__name__``; ``build_from_cfg`` pops ``type``, looks it up
(``KeyError`` if unknown, ``TypeError`` if not a str/class) and calls ``cls(**kwargs, cfg=cfg)``;
registering a duplicate name raises ``KeyError``.
"""
import inspect

import torch.nn as nn


class Registry:
    def __init__(self, name):
        self._name = name
        self._module_dict = {}

    def __repr__(self):
        return f'{type(self).__name__}(name={self._name}, items={list(self._module_dict)})'

    @property
    def name(self):
        return self._name

    @property
    def module_dict(self):
        return self._module_dict

    def get(self, key):
        return self._module_dict.get(key)

    def register_module(self, cls):
        if not inspect.isclass(cls):
            raise TypeError(f'module must be a class, but got {type(cls)}')
        if cls.__name__ in self._module_dict:
            raise KeyError(f'{cls.__name__} is already registered in {self._name}')
        self._module_dict[cls.__name__] = cls
        return cls


def build_from_cfg(cfg, registry, default_args=None):
    assert isinstance(cfg, dict) and 'type' in cfg
    assert isinstance(default_args, dict) or default_args is None
    args = dict(cfg)
    kind = args.pop('type')
    if isinstance(kind, str):
        cls = registry.get(kind)
        if cls is None:
            raise KeyError(f'{kind} is not in the {registry.name} registry')
    elif inspect.isclass(kind):
        cls = kind
    else:
        raise TypeError(f'type must be a str or valid type, but got {type(kind)}')
    for k, v in (default_args or {}).items():
        args.setdefault(k, v)
    return cls(**args)


PCENCODER = Registry('pcencoder')
BACKBONE = Registry('backbone')
HEADS = Registry('heads')
NET = Registry('net')
BKDECODER = Registry('bkdecoder')


def build(cfg, registry, default_args=None):
    if isinstance(cfg, list):
        return nn.Sequential(*[build_from_cfg(c, registry, default_args) for c in cfg])
    return build_from_cfg(cfg, registry, default_args)


def build_pcencoder(cfg):
    return build(cfg.pcencoder, PCENCODER, default_args=dict(cfg=cfg))


def build_backbone(cfg):
    return build(cfg.backbone, BACKBONE, default_args=dict(cfg=cfg))


def build_heads(cfg):
    return build(cfg.heads, HEADS, default_args=dict(cfg=cfg))


def build_net(cfg):
    return build(cfg.net, NET, default_args=dict(cfg=cfg))
