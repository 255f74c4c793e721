#!/usr/bin/env python3
"""Split-precision (bf16x3) implicit Winograd kernel alone on the FPN's shapes: ms, fp32-equivalent TFLOP/s, max |diff| vs the fp32 kernel."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lanemapping_amd import ops
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
out = []
for cin, cout, dil, hw in [(256, 256, 1, 288), (256, 128, 1, 288), (256, 256, 2, 144), (128, 128, 1, 144), (128, 256, 1, 144), (64, 64, 1, 288)]:
    x = ops.new_act(B, cin, hw, hw, dev).normal_()
    w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    wu = ops.pack_wino(w)
    wf, wf3 = ops.pack_wino_fragments(wu), ops.pack_wino_fragments_bf16x3(wu)
    y32 = ops.conv_wino_implicit(x, wf, cout, dil)
    y = ops.conv_wino_implicit(x, wf3, cout, dil)
    err = float((y - y32).abs().max())
    res = {}
    import ctypes
    has_prof = hasattr(ops.lib(), 'lm_iprof_read')
    for name, f in (('fp32', wf), ('bf16x3', wf3)):
        for _ in range(2):
            ops.conv_wino_implicit(x, f, cout, dil, out=y)
        torch.cuda.synchronize()
        if has_prof:
            buf = (ctypes.c_ulonglong * 12)()
            ops.lib().lm_iprof_read(buf, 1)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ops.conv_wino_implicit(x, f, cout, dil, out=y)
        b.record()
        torch.cuda.synchronize()
        res[name] = a.elapsed_time(b) / 10
        if has_prof and name == 'bf16x3':
            ops.lib().lm_iprof_read(buf, 1)
            nw = max(buf[11], 1)
            print(f'  iprof {cin}->{cout}@{hw}: ' + ' '.join(f'{k}={buf[k] / nw:.0f}' for k in range(11)) + f' total={sum(buf[:11]) / nw:.0f}')
    out.append(f'{cin}->{cout} d{dil}@{hw} fp32 {res["fp32"]:.3f} split {res["bf16x3"]:.3f} ms (x{res["fp32"] / res["bf16x3"]:.2f}) err {err:.1e}')
print(os.path.basename(os.environ.get('LANEMAP_HIP_LIB', 'product')), ' | '.join(out))
