"""Helpers shared by the GPU parity test files (tests/test_gpu_*.py)."""
import os

import numpy as np
import torch

from lanemapping_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _close(a, ref, tol=1e-4, name=''):
    a = a.detach().float().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
    ref = ref.detach().float().cpu().numpy() if torch.is_tensor(ref) else np.asarray(ref)
    assert a.shape == ref.shape, (name, a.shape, ref.shape)
    scale = max(1.0, float(np.abs(ref).max()))
    err = float(np.abs(a - ref).max())
    assert err <= tol * scale, f'{name}: max err {err:.3e} > {tol:.0e} * scale {scale:.3f}'
    return err


def _rowref_head(dev):
    from lanemapping_amd.boundary import load_config
    from lanemapping_amd.registry import build_heads
    cfg = load_config('Proj28_GFC-T3_RowRef_82_73_laser')
    head = build_heads(cfg).eval()
    synth.fill_module_(head, 2021, prefix='heads.')

    class Emb(torch.nn.Module):       # the reference keeps emb_c as Parameters under the CPU stub: same name-keyed values
        def __init__(self):
            super().__init__()
            for c in range(12):
                setattr(self, f'emb_{c}', torch.nn.Parameter(torch.zeros(1024)))
    e = synth.fill_module_(Emb(), 2021, prefix='heads.')
    head.set_lane_embeddings([getattr(e, f'emb_{c}').detach() for c in range(12)])
    return head.to(dev)


# ----------------------------------------------------------------------------------------------- config 5 (LiDAR encoder)
# voxeliser + sparse convolutions: PARITY UNPINNED (third-party arithmetic, oracle = restated published behaviour);
# dense tail: pinned by G11 (generated from the reference).
def _lidar_module(dev, cfg, seed=2021):
    from lanemapping_amd import lidarencoder  # noqa: F401
    from lanemapping_amd.registry import build_pcencoder
    m = build_pcencoder(cfg).eval()
    synth.fill_module_(m, seed, prefix='pcencoder.')
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    return m.to(dev), sd
