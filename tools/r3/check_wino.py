#!/usr/bin/env python3
"""Bit-identity of the implicit Winograd launcher's choice against the materialising pair on ragged shapes (env switches pick the geometry)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lanemapping_amd import ops
dev = torch.device('cuda:0')
bad = 0
for (B, cin, cout, H, W, dil) in [(2, 128, 128, 84, 90, 2), (1, 256, 200, 43, 61, 1), (2, 64, 128, 96, 100, 1), (1, 160, 256, 85, 87, 2),
                                  (2, 256, 256, 144, 144, 1), (1, 256, 512, 72, 72, 1), (3, 32, 96, 150, 140, 3), (1, 128, 256, 288, 288, 1)]:
    if not ops.wino_implicit_supported(H, W, cin, dil):
        print('skip', (B, cin, cout, H, W, dil))
        continue
    g = torch.Generator().manual_seed(cin + cout + H)
    x = torch.randn((B, cin, H, W), generator=g).to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    r = torch.randn((B, cout, H, W), generator=g).to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    w = (torch.randn((cout, cin, 3, 3), generator=g) / (cin * 9) ** 0.5).to(dev)
    sc, sh = (torch.rand(cout, generator=g) + 0.5).to(dev), torch.randn(cout, generator=g).to(dev)
    wu = ops.pack_wino(w)
    wf = ops.pack_wino_fragments(wu)
    y0 = ops.conv_wino(x, wu, cout, dil, scale=sc, shift=sh, res=r, act=ops.ACT_RELU)
    for rep in range(3):
        y1 = ops.conv_wino_implicit(x, wf, cout, dil, scale=sc, shift=sh, res=r, act=ops.ACT_RELU)
        if not torch.equal(y0, y1):
            d = (y0 - y1).abs()
            print('MISMATCH', (B, cin, cout, H, W, dil), 'rep', rep, 'max', float(d.max()), 'count', int((d > 0).sum()), 'of', d.numel())
            bad += 1
            break
    if cout % 4 == 0:
        a, sa = ops.conv_wino(x, wu, cout, dil, shift=sh, gn_eps=1e-5)
        b, sb = ops.conv_wino_implicit(x, wf, cout, dil, shift=sh, gn_eps=1e-5)
        if not (torch.equal(a, b) and torch.allclose(sa, sb, rtol=2e-5, atol=1e-6)):
            print('GN MISMATCH', (B, cin, cout, H, W, dil), float((a - b).abs().max()), float((sa - sb).abs().max()))
            bad += 1
print({k: v for k, v in os.environ.items() if k.startswith('LANEMAP_WINO')}, 'ok' if bad == 0 else f'{bad} FAILED')
