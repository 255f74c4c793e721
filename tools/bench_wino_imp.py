#!/usr/bin/env python3
"""Implicit-transform Winograd kernel alone on the FPN's big shapes: ms and executed TFLOP/s (timing ablation builds: LANEMAP_HIP_LIB).
BF16X3=1: the split-precision variant."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanemapping_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SPLIT = os.environ.get('BF16X3', '0') != '0'
SHAPES = [(256, 256, 1, 288), (256, 256, 2, 144), (128, 128, 1, 144)]
res = []
for cin, cout, dil, hw in SHAPES:
    x = ops.new_act(B, cin, hw, hw, dev).normal_()
    w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    wu = ops.pack_wino(w)
    wf = ops.pack_wino_fragments_bf16x3(wu) if SPLIT else ops.pack_wino_fragments(wu)
    y = ops.conv_wino_implicit(x, wf, cout, dil)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        ops.conv_wino_implicit(x, wf, cout, dil, out=y)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    if hasattr(ops.lib(), 'lm_iprof_read'):     # LM_IPROF probe build: mean shader-clock cycles per wave and phase
        import ctypes
        buf = (ctypes.c_ulonglong * 12)()
        ops.lib().lm_iprof_read(buf, 1)
        nw = max(buf[11], 1)
        names = ['prologue', 'transform', 'barrier1', 'mfma', 'slabwait', 'barrier2', 'epi_tail', 'epi_setup', 'epi_fold', 'epi_xpose', 'epi_store']
        print(f'  iprof {cin}->{cout} d{dil} @{hw}: ' + ' '.join(f'{n}={buf[i] / nw:.0f}' for i, n in enumerate(names)) +
              f' total={sum(buf[:11]) / nw:.0f} cycles/wave')
    tiles = ops.lib().lm_conv3x3_winograd_workspace_bytes(B, hw, hw, cin, dil) // (64 * cin)
    res.append(f'{cin}->{cout} d{dil} @{hw}: {ms:.3f} ms {2.0 * 16 * tiles * cin * cout / ms / 1e9:6.1f} TF(fp32-eq)')
print(os.environ.get('LANEMAP_HIP_LIB', 'product'), 'bf16x3' if SPLIT else 'fp32', ' | '.join(res))
