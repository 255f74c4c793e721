#!/usr/bin/env python3
"""Implicit-transform Winograd kernel alone on the FPN's big shapes: ms and executed TFLOP/s (timing ablation builds: LANEMAP_HIP_LIB)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanemapping_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SHAPES = [(256, 256, 1, 288), (256, 256, 2, 144), (128, 128, 1, 144)]
res = []
for cin, cout, dil, hw in SHAPES:
    x = ops.new_act(B, cin, hw, hw, dev).normal_()
    w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    wf = ops.pack_wino_fragments(ops.pack_wino(w))
    y = ops.conv_wino_implicit(x, wf, cout, dil)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        ops.conv_wino_implicit(x, wf, cout, dil, out=y)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    tiles = ops.lib().lm_conv3x3_winograd_workspace_bytes(B, hw, hw, cin, dil) // (64 * cin)
    res.append(f'{cin}->{cout} d{dil} @{hw}: {ms:.3f} ms {2.0 * 16 * tiles * cin * cout / ms / 1e9:6.1f} TF')
print(os.environ.get('LANEMAP_HIP_LIB', 'product'), ' | '.join(res))
