"""Import harness for the READ-ONLY upstream reference at /root/reference.

Used ONLY by tests/golden/make_golden.py inside the build container to produce
golden vectors.  Nothing here (and nothing from /root/reference) travels to the
GPU box: the committed artefacts are the .npz fixtures this harness helps emit.

Recipe = SURVEY.md Appendix B: absent third-party packages are replaced by
MagicMock modules, `addict.Dict` by a 30-line attribute dict, `.cuda()` by the
identity so the reference runs on CPU.
"""
import importlib.abc
import importlib.machinery
import sys
import types
from unittest import mock

REF_ROOT = '/root/reference'
_STUB_ROOTS = {'cv2', 'torchvision', 'skimage', 'mmcv', 'timm', 'mmdet3d', 'mmengine',
               'laspy', 'open3d', 'yapf', 'pytorch_warmup', 'tensorboard', 'tensorboardX'}


class _StubLoader(importlib.abc.Loader):
    def create_module(self, spec):
        m = mock.MagicMock(name=spec.name)
        m.__name__ = spec.name
        m.__path__ = []
        m.__spec__ = spec
        m.__loader__ = self
        return m

    def exec_module(self, module):
        pass


class _StubFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, name, path=None, target=None):
        if name.split('.')[0] in _STUB_ROOTS:
            return importlib.machinery.ModuleSpec(name, _StubLoader(), is_package=True)
        return None


class _AttrDict(dict):
    """Minimal stand-in for addict.Dict (recursive attribute access)."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        for k, v in dict(*args, **kwargs).items():
            self[k] = self._hook(v)

    @classmethod
    def _hook(cls, v):
        if isinstance(v, dict) and not isinstance(v, cls):
            return cls(v)
        if isinstance(v, (list, tuple)):
            return type(v)(cls._hook(i) for i in v)
        return v

    def __missing__(self, name):
        raise KeyError(name)

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = self._hook(value)

    def __setitem__(self, name, value):
        super().__setitem__(name, self._hook(value))

    def copy(self):
        return type(self)(self)

    def to_dict(self):
        return {k: (v.to_dict() if isinstance(v, _AttrDict) else v) for k, v in self.items()}


def install():
    """Make `import baseline...` work on CPU in this container."""
    import torch
    if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
        sys.meta_path.insert(0, _StubFinder())
    addict = types.ModuleType('addict')
    addict.Dict = _AttrDict
    sys.modules.setdefault('addict', addict)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)


def load_cfg(rel_path, **overrides):
    install()
    from baseline.utils.config import Config
    cfg = Config.fromfile(f'{REF_ROOT}/{rel_path}')
    cfg.pcencoder.pretrained = False
    for k, v in overrides.items():
        setattr(cfg, k, v)
    return cfg


def build_ref_net(cfg, seed=2021):
    import torch
    from baseline.models.registry import build_net
    torch.manual_seed(seed)
    net = build_net(cfg).eval()
    return net
