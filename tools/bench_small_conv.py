#!/usr/bin/env python3
"""Micro-bench of lm_conv2d_nhwc_small on the FPN's 1x1 output layers (B = 8)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanemapping_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
B = 8
for cin, cout, hw, k in ((128, 8, 288, 1), (128, 1, 288, 1), (8, 3, 288, 1), (16, 16, 144, 3)):
    x = ops.new_act(B, cin, hw, hw, dev).normal_()
    w = torch.randn(cout, cin, k, k, device=dev)
    wp = ops.pack_small(w)
    sh = torch.randn(cout, device=dev)
    fn = lambda: ops.conv_small(x, wp, cout, shift=sh, **({'kh': k, 'kw': k, 'pad': k // 2} if k > 1 else {}))
    try:
        fn()
    except TypeError:
        fn = lambda: ops.conv_small(x, wp, cout, shift=sh)
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
    gb = B * hw * hw * (cin + cout) * 4 / 1e9
    print(f'{cin}->{cout} k{k} @{hw}: {ms:.3f} ms  {gb / ms * 1e3:.0f} GB/s')
