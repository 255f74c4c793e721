#!/usr/bin/env python3
"""Winograd F(2x2,3x3) vs the direct MFMA convolution on the FPN's 3x3 shapes: max |diff| and (direct-equivalent) TFLOP/s."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanemapping_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
SHAPES = [(256, 256, 2, 144), (256, 128, 1, 288), (256, 256, 1, 288), (256, 256, 1, 144), (128, 128, 1, 144), (64, 64, 1, 288), (128, 256, 1, 144)]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for cin, cout, dil, hw in SHAPES:
    x = ops.new_act(B, cin, hw, hw, dev).normal_()
    w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    shift = torch.randn(cout, device=dev)
    res = ops.new_act(B, cout, hw, hw, dev).normal_()
    wp, wu = ops.pack_mfma(w), ops.pack_wino(w)
    wf = ops.pack_wino_fragments(wu)
    wf3 = ops.pack_wino_fragments_bf16x3(wu)
    imp = ops.wino_implicit_supported(hw, hw, cin, dil)
    yd = ops.conv_mfma(x, wp, cout, 3, 3, 1, dil, dil, shift=shift, res=res, act=ops.ACT_RELU)
    yw = ops.conv_wino(x, wu, cout, dil, shift=shift, res=res, act=ops.ACT_RELU)
    err = float((yd - yw).abs().max())
    out = {}
    same = None
    cases = [('direct', lambda: ops.conv_mfma(x, wp, cout, 3, 3, 1, dil, dil, shift=shift, out=yd)),
             ('wino', lambda: ops.conv_wino(x, wu, cout, dil, shift=shift, out=yw))]
    if imp:
        yi = ops.conv_wino_implicit(x, wf, cout, dil, shift=shift, res=res, act=ops.ACT_RELU)
        same = bool(torch.equal(yi, yw))
        cases.append(('implicit', lambda: ops.conv_wino_implicit(x, wf, cout, dil, shift=shift, out=yi)))
        ys = ops.conv_wino_implicit(x, wf3, cout, dil, shift=shift, res=res, act=ops.ACT_RELU)
        err3 = float((ys - yw).abs().max())
        cases.append(('bf16x3', lambda: ops.conv_wino_implicit(x, wf3, cout, dil, shift=shift, out=ys)))
    for name, fn in cases:
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            fn()
        b.record()
        torch.cuda.synchronize()
        out[name] = a.elapsed_time(b) / 10
    fl = 2.0 * B * hw * hw * cout * cin * 9
    print(f'{cin:4d}->{cout:4d} d{dil} @{hw}: max|diff| {err:.2e}  direct {out["direct"]:.3f} ms ({fl / out["direct"] / 1e9:6.1f} TF)  '
          f'wino {out["wino"]:.3f} ms ({fl / out["wino"] / 1e9:6.1f} TF-equivalent)  x{out["direct"] / out["wino"]:.2f}'
          + (f'  implicit {out["implicit"]:.3f} ms ({fl / out["implicit"] / 1e9:6.1f} TF-eq, {fl * 16 / 36 / out["implicit"] / 1e9:6.1f} executed)  '
             f'bit-identical to wino: {same}  |  bf16x3 {out["bf16x3"]:.3f} ms ({fl / out["bf16x3"] / 1e9:6.1f} TF-eq) max|diff| vs fp32 wino {err3:.2e}' if imp else ''))
