#!/usr/bin/env python3
"""Every MFMA convolution launch of one network forward (config argv[2] = config 2, batch argv[1] = 4): label, ms, executed TFLOP/s, sorted by time."""
import os
import sys
import collections

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanemapping_amd import ops, synth  # noqa: E402
from lanemapping_amd.boundary import build_net_from_config  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
CFG = sys.argv[2] if len(sys.argv) > 2 else 'Proj_polyline_fpn_vit_vertex_2'
dev = torch.device('cuda:0')
net = build_net_from_config(CFG, device='cpu')
synth.fill_module_(net, 2021)
net = net.to(dev)
x = torch.from_numpy(synth.bev_batch([2021 + i for i in range(B)], 1152)).to(dev)
rec = collections.OrderedDict()
on = [False]


def hook(kind, flops, launch, executed=None):
    if not on[0]:
        return launch()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    launch()
    b.record()
    rec.setdefault(kind, []).append((a, b, flops if executed is None else executed))


ops.set_conv_hook(hook)
with torch.no_grad():
    net({'proj': x})
    on[0] = True
    for _ in range(3):
        net({'proj': x})
torch.cuda.synchronize()
rows = []
for k, v in rec.items():
    ms = sum(a.elapsed_time(b) for a, b, _ in v) / 3
    rows.append((ms, k, len(v) // 3, sum(f for _, _, f in v) / 3))
tot = sum(r[0] for r in rows)
for ms, k, n, f in sorted(rows, reverse=True):
    print(f'{ms:8.3f} ms {100 * ms / tot:5.1f}%  x{n:2d}  {f / ms / 1e9:6.1f} TF  {k}')
print(f'{tot:8.3f} ms total, batch {B}')
