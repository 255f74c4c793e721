#!/usr/bin/env python3
"""Implicit Winograd launches WITH the BasicBlock epilogue (BN scale / shift, residual, ReLU) on the FPN's shapes: ms per launch."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lanemapping_amd import ops
dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
out = []
for cin, cout, dil, hw in [(256, 256, 1, 288), (256, 256, 2, 144), (128, 128, 1, 144), (256, 128, 1, 288)]:
    x = ops.new_act(B, cin, hw, hw, dev).normal_()
    r = ops.new_act(B, cout, hw, hw, dev).normal_()
    w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    wu = ops.pack_wino(w)
    wf = ops.pack_wino_fragments(wu)
    y = ops.conv_wino_implicit(x, wf, cout, dil, scale=sc, shift=sh, res=r, act=ops.ACT_RELU)
    y0 = ops.conv_wino(x, wu, cout, dil, scale=sc, shift=sh, res=r, act=ops.ACT_RELU)
    ok = torch.equal(y, y0)
    del y0
    res = {}
    for name, kw in (('plain', {}), ('bn+res+relu', dict(scale=sc, shift=sh, res=r, act=ops.ACT_RELU))):
        for _ in range(2):
            ops.conv_wino_implicit(x, wf, cout, dil, out=y, **kw)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            ops.conv_wino_implicit(x, wf, cout, dil, out=y, **kw)
        b.record()
        torch.cuda.synchronize()
        res[name] = a.elapsed_time(b) / 10
    out.append(f'{cin}->{cout} d{dil}@{hw} plain {res["plain"]:.3f} / res {res["bn+res+relu"]:.3f} ms{"" if ok else " MISMATCH"}')
print(os.path.basename(os.environ.get('LANEMAP_HIP_LIB', 'product')), ' | '.join(out))
