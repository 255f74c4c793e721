"""Convenience entry points of the drop-in boundary: load a `configs/Proj_*.py`, build the net through
the registries (importing this module registers every hot-path class), load a reference checkpoint."""
import os

import torch

from .config import Config, apply_inference_defaults
from .registry import build_net, PCENCODER, BACKBONE, HEADS, NET   # noqa: F401
from . import pcencoder, lidarencoder, backbone, heads, rowref, net   # noqa: F401  (registration side effects)

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_config(path_or_name):
    path = path_or_name
    if not os.path.isfile(path):
        path = os.path.join(REPO_ROOT, 'configs', path_or_name + ('' if path_or_name.endswith('.py') else '.py'))
    return apply_inference_defaults(Config.fromfile(path))


def build_net_from_config(path_or_name, device='cuda', **overrides):
    cfg = load_config(path_or_name)
    for k, v in overrides.items():
        cfg[k] = v
    model = build_net(cfg).eval()
    return model.to(device)


def load_reference_checkpoint(model, path, strict=True):
    """Reference checkpoints are {'net': state_dict, ...} saved from nn.DataParallel, i.e. keys carry a
    `module.` prefix (engine/runner.py:103-104,399-401; utils/net_utils.py:35-45)."""
    ckpt = torch.load(path, map_location='cpu')
    sd = ckpt['net'] if isinstance(ckpt, dict) and 'net' in ckpt else ckpt
    sd = {(k[len('module.'):] if k.startswith('module.') else k): v for k, v in sd.items()}
    return model.load_state_dict(sd, strict=strict)
