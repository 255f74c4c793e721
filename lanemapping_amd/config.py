"""Python-file configs (`configs/Proj_*.py`) without addict/yapf.

Mirrors what the hot path needs from the reference's mmcv-style loader (baseline/utils/config.py:22-36,
88-123): the file is executed, its public globals become a ``ConfigDict`` with attribute access, nested
dicts are wrapped too, and a missing key raises ``AttributeError``.  ``apply_inference_defaults`` injects
the keys the published configs forget.
"""
import os
import types


class ConfigDict(dict):
    def __init__(self, *a, **k):
        super().__init__()
        for key, v in dict(*a, **k).items():
            self[key] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, ConfigDict):
            return ConfigDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(ConfigDict._wrap(i) for i in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(f"'ConfigDict' object has no attribute '{name}'")

    def __setattr__(self, name, value):
        self[name] = value

    def copy(self):
        return ConfigDict(self)


class Config(ConfigDict):
    @staticmethod
    def fromfile(path):
        path = os.path.abspath(os.path.expanduser(path))
        if not os.path.isfile(path):
            raise FileNotFoundError(f'file "{path}" does not exist')
        if not path.endswith('.py'):
            raise IOError('Only py type are supported now!')
        scope = {'__file__': path, '__name__': '_lanemap_cfg'}
        with open(path) as f:
            exec(compile(f.read(), path, 'exec'), scope)
        cfg = Config({k: v for k, v in scope.items()
                      if not k.startswith('__') and not isinstance(v, (types.ModuleType, types.FunctionType, type))})
        cfg['filename'] = path
        return cfg


_DEFAULTS = dict(vit_seg=True, is_gt_avai=False, view=False, view_detail=False, show_result=False, column_att=False,
                 column_transformer_decoder=False, spatial_att=True, cls_smooth=False, validate_buffer=10, gt_downsample_ratio=8,
                 flip_label=False, number_lanes=12, number_orients=11, dataset_type='LaserLaneProposal',
                 proposal_obj_thre=0.3, exist_thre=0.2, coor_thre=0.2, endp_thre=0.08, seg_thre=0.1,
                 featuremap_out_channel=64)


def apply_inference_defaults(cfg):
    for k, v in _DEFAULTS.items():
        if k not in cfg:
            cfg[k] = v
    if 'pcencoder' in cfg and 'pretrained' in cfg.pcencoder:
        cfg.pcencoder['pretrained'] = False   # no network: weights always come from a checkpoint
    return cfg
