#!/usr/bin/env python3
"""Micro-bench of lm_conv2d_nhwc_mfma_f32 on the FPN's dominant shapes, per tile-selection variant."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanemapping_amd import ops  # noqa: E402
from lanemapping_amd._lib import lib  # noqa: E402

dev = torch.device('cuda:0')
SHAPES = [(256, 256, 3, 1, 2, 144), (256, 128, 3, 1, 1, 288), (256, 256, 3, 1, 1, 288), (128, 128, 3, 1, 1, 144),
          (64, 64, 3, 1, 1, 288), (64, 256, 1, 1, 1, 288)]
variants = [int(v) for v in sys.argv[1:]] or [0]
B = 8
for cin, cout, k, stride, dil, hw in SHAPES:
    x = ops.new_act(B, cin, hw, hw, dev).normal_()
    w = ops.pack_mfma(torch.randn(cout, cin, k, k, device=dev) / (cin * k * k) ** 0.5)
    y = ops.new_act(B, cout, hw, hw, dev)
    res = {}
    for v in variants:
        for _ in range(2):
            ops.conv_mfma(x, w, cout, k, k, stride, dil * (k // 2), dil, out=y)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        n = 10
        for _ in range(n):
            ops.conv_mfma(x, w, cout, k, k, stride, dil * (k // 2), dil, out=y)
        b.record()
        torch.cuda.synchronize()
        ms = a.elapsed_time(b) / n
        res[v] = 2.0 * B * hw * hw * cout * cin * k * k / ms / 1e9
    print(f'{cin:4d}->{cout:4d} k{k} d{dil} @{hw}: ' + '  '.join(f'v{v}: {t:6.1f} TF' for v, t in res.items()))
