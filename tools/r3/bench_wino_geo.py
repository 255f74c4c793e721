#!/usr/bin/env python3
"""Implicit Winograd kernel alone on the FPN's big shapes (env switches pick the geometry, LANEMAP_HIP_LIB a variant build); on an
LM_IPROF build the phase cycles of wave 0 per workgroup."""
import ctypes
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lanemapping_amd import ops
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
SHAPES = [(256, 256, 1, 288), (256, 128, 1, 288), (256, 256, 2, 144), (128, 128, 1, 144), (128, 256, 1, 144), (256, 512, 1, 144)][:int(os.environ.get('NSHAPES', '6'))]
out = []
for cin, cout, dil, hw in SHAPES:
    x = ops.new_act(B, cin, hw, hw, dev).normal_()
    w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    wu = ops.pack_wino(w)
    wf = ops.pack_wino_fragments(wu)
    y = ops.conv_wino_implicit(x, wf, cout, dil)
    if os.environ.get('CHECK', '1') != '0':
        y0 = ops.conv_wino(x, wu, cout, dil)
        if not torch.equal(y, y0):
            print('   MISMATCH vs the materialising pair', cin, cout, dil, hw, float((y - y0).abs().max()))
        del y0
    torch.cuda.synchronize()
    has_prof = hasattr(ops.lib(), 'lm_iprof_read')
    if has_prof:
        buf = (ctypes.c_ulonglong * 12)()
        ops.lib().lm_iprof_read(buf, 1)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        ops.conv_wino_implicit(x, wf, cout, dil, out=y)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    if has_prof:
        ops.lib().lm_iprof_read(buf, 1)
        nw = max(buf[11], 1)
        names = ['prologue', 'transform', 'barrier1', 'mfma', 'slabwait', 'barrier2', 'epilogue', 'setup']
        print(f'  iprof {cin}->{cout} d{dil} @{hw}: ' + ' '.join(f'{n}={buf[i] / nw:.0f}' for i, n in enumerate(names)) + f' total={sum(buf[:11]) / nw:.0f} cycles/workgroup')
    tiles = ops.lib().lm_conv3x3_winograd_workspace_bytes(B, hw, hw, cin, dil) // (64 * cin)
    out.append(f'{cin}->{cout} d{dil}@{hw} {ms:.3f} ms {2.0 * 16 * tiles * cin * cout / ms / 1e9:5.1f} TF')
print(os.path.basename(os.environ.get('LANEMAP_HIP_LIB', 'product')), {k[13:]: v for k, v in os.environ.items() if k.startswith('LANEMAP_WINO')}, ' | '.join(out))
