#!/usr/bin/env python3
"""ATen GPU kernels (copies, fills, element-wise) launched inside one steady-state batch of the tile pipeline, with the Python line that
launched them (torch.profiler with stacks: the stage custom ops hide their inner calls from a TorchDispatchMode)."""
import collections
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lanemapping_amd import synth  # noqa: E402
from lanemapping_amd.boundary import build_net_from_config  # noqa: E402
from lanemapping_amd.pipeline import TilePipeline  # noqa: E402

dev = torch.device('cuda:0')
net = build_net_from_config(sys.argv[1] if len(sys.argv) > 1 else 'Proj_polyline_fpn_vit_vertex_2', device='cpu')
synth.fill_module_(net, 2021)
net = net.to(dev)
x = torch.from_numpy(synth.bev_batch([1, 2], 1152)).to(dev)
pipe = TilePipeline(net)
for _ in range(2):
    pipe.run_batch(x)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    pipe.run_batch(x)
    torch.cuda.synchronize()
rows = collections.Counter()
for ev in prof.events():
    if not ev.name.startswith('aten::') or ev.device_time_total <= 0 and not ev.kernels:
        continue
    if not ev.kernels:
        continue
    where = next((s for s in ev.stack if 'lanemapping_amd' in s), ev.stack[0] if ev.stack else '?')
    rows[(ev.name, where.strip()[-90:])] += 1
for (name, where), n in rows.most_common(40):
    print(f'{n:4d}  {name:28s} {where}')
