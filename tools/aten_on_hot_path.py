#!/usr/bin/env python3
"""Which ATen ops (kernels, not views) run inside one steady-state batch of the tile pipeline, and from which line of the package."""
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanemapping_amd import synth  # noqa: E402
from lanemapping_amd.boundary import build_net_from_config  # noqa: E402
from lanemapping_amd.pipeline import TilePipeline  # noqa: E402

VIEWS = {'empty', 'empty_strided', 'permute', 'view', 'reshape', 'as_strided', 'slice', 'select', '_unsafe_view', 'empty_like', 'expand',
         'transpose', 't', 'alias', 'detach', 'squeeze', 'unsqueeze', 'narrow', 'is_pinned', '_local_scalar_dense', 'lift_fresh',
         'record_stream', 'is_same_size', 'sym_size', 'sym_stride', 'stride', 'size'}
rows = {}


class Log(TorchDispatchMode):
    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = func.__name__.split('.')[0]
        if name not in VIEWS:
            fr = [f for f in traceback.extract_stack() if 'lanemapping_amd' in f.filename]
            where = f'{os.path.basename(fr[-1].filename)}:{fr[-1].lineno}' if fr else '?'
            dev = next((str(a.device) for a in args if isinstance(a, torch.Tensor)), '')
            k = (name, where, dev)
            rows[k] = rows.get(k, 0) + 1
        return func(*args, **(kwargs or {}))


dev = torch.device('cuda:0')
net = build_net_from_config(sys.argv[1] if len(sys.argv) > 1 else 'Proj_polyline_fpn_vit_vertex_2', device='cpu')
synth.fill_module_(net, 2021)
net = net.to(dev)
x = torch.from_numpy(synth.bev_batch([1, 2], 1152)).to(dev)
pipe = TilePipeline(net)
for _ in range(2):
    pipe.run_batch(x)
torch.cuda.synchronize()
with Log():
    pipe.run_batch(x)
torch.cuda.synchronize()
for (name, where, d), n in sorted(rows.items(), key=lambda kv: -kv[1]):
    print(f'{n:4d}  {name:24s} {d:8s} {where}')
