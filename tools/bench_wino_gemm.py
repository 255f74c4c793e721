#!/usr/bin/env python3
"""Times ONLY the Winograd batched GEMM (lm_winograd_gemm_f32) on a pre-transformed input: executed TFLOP/s of the matrix cores.
usage: bench_wino_gemm.py [cin cout hw dil B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanemapping_amd import ops  # noqa: E402

a = [int(v) for v in sys.argv[1:]]
cin, cout, hw, dil, B = (a + [256, 256, 288, 1, 8][len(a):])
dev = torch.device('cuda:0')
x = ops.new_act(B, cin, hw, hw, dev).normal_()
w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
wu = ops.pack_wino(w)
v = ops.wino_transform(x, dil, dedicated=True)
y = ops.new_act(B, cout, hw, hw, dev)
for _ in range(3):
    ops.conv_wino(v, wu, cout, dil, out=y)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    ops.conv_wino(v, wu, cout, dil, out=y)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
tiles = B * ((hw // dil + 1) // 2) ** 2 * dil * dil
fl = 2.0 * 16 * tiles * cin * cout
print(f'{cin}->{cout} @{hw} d{dil} B{B}: gemm {ms:.3f} ms  {fl / ms / 1e9:.1f} TFLOP/s executed (approx. tile count)')
