import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name))
    return load


@pytest.fixture(scope='session')
def synth_sd():
    """Synthetic state dict (reference key names) for config 2, weight seed 2021."""
    from lanemapping_amd import synth
    from lanemapping_amd.boundary import build_net_from_config
    net = build_net_from_config('Proj_polyline_fpn_vit_vertex_2', device='cpu')
    synth.fill_module_(net, 2021)
    return {k: v.clone() for k, v in net.state_dict().items()}


@pytest.fixture(scope='session')
def dev():
    import torch
    assert torch.cuda.is_available(), 'these tests need an MI355X'
    from lanemapping_amd._lib import lib
    lib()   # fail loudly if the HIP library is missing
    return torch.device('cuda:0')


@pytest.fixture(scope='session')
def net(dev):
    """Config-2 detector with the synthetic weights of seed 2021 on the GPU (shared by the test_gpu_* files; tests do not mutate it)."""
    from lanemapping_amd import synth
    from lanemapping_amd.boundary import build_net_from_config
    n = build_net_from_config('Proj_polyline_fpn_vit_vertex_2', device='cpu')
    synth.fill_module_(n, 2021)
    return n.to(dev)
