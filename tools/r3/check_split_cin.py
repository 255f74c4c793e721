#!/usr/bin/env python3
"""Split-precision implicit Winograd kernel on the slot counts the FPN does not have (Cin = 32: two slots, 96: six) and a 32-channel
output: max |diff| vs the fp32 kernel and both vs the fp64 convolution."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lanemapping_amd import ops
dev = torch.device('cuda:0')
for (B, cin, cout, H, W, dil) in [(1, 32, 64, 100, 96, 1), (2, 96, 128, 60, 90, 1), (1, 32, 32, 144, 144, 2), (1, 64, 256, 288, 288, 1)]:
    x = ops.new_act(B, cin, H, W, dev).normal_()
    w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    wu = ops.pack_wino(w)
    y32 = ops.conv_wino_implicit(x, ops.pack_wino_fragments(wu), cout, dil)
    y3 = ops.conv_wino_implicit(x, ops.pack_wino_fragments_bf16x3(wu), cout, dil)
    ref = torch.nn.functional.conv2d(x.double(), w.double(), None, 1, dil, dil).float()
    print(cin, cout, H, W, dil, 'split-fp32', float((y3 - y32).abs().max()), 'split-fp64', float((y3 - ref).abs().max()), 'fp32-fp64', float((y32 - ref).abs().max()))
