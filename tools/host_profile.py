#!/usr/bin/env python3
"""cProfile of the host-side enqueue of one sub-batch (2 tiles): where do the ~4.6 ms of Python go?"""
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanemapping_amd import synth  # noqa: E402
from lanemapping_amd.boundary import build_net_from_config  # noqa: E402
from lanemapping_amd.pipeline import TilePipeline  # noqa: E402

dev = torch.device('cuda:0')
net = build_net_from_config('Proj_polyline_fpn_vit_vertex_2', device='cpu')
synth.fill_module_(net, 2021)
net = net.to(dev)
tiles = torch.from_numpy(synth.bev_batch([2021, 2022], 1152)).to(dev)
pipe = TilePipeline(net)
with torch.no_grad():
    for _ in range(3):
        pipe._gpu_stage(tiles)
    torch.cuda.synchronize()
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(20):
        pipe._gpu_stage(tiles)
    pr.disable()
    torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(22)
