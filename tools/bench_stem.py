#!/usr/bin/env python3
"""Micro-bench of the 7x7/s2 stem + BN + ReLU (3 -> 64 channels) and the 3x3/s2 max-pool on 8 tiles of 1152^2."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanemapping_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
B = 8
x = torch.rand(B, 3, 1152, 1152, device=dev)
w = torch.randn(7, 7, 3, 64, device=dev) * 0.1
s, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev)
for name, fn in (('stem', lambda: ops.stem(x, w, s, sh)), ('stem+pool', lambda: ops.maxpool3x3s2(ops.stem(x, w, s, sh)))):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    print(f'{name}: {ms:.3f} ms  ({2.0 * B * 576 * 576 * 64 * 147 / ms / 1e9:.1f} TFLOP/s VALU)')
