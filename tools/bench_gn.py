#!/usr/bin/env python3
"""Micro-bench of lm_gn_relu_upsample (64-bit index math) vs lm_gn_relu_upsample_sum with one term (32-bit) on the FPN's shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanemapping_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
B=8
for (C,hi,ho) in ((256,144,288),(128,288,288),(128,144,288)):
    x = ops.new_act(B,C,hi,hi,dev).normal_()
    st = ops.gn_stats(x)
    g = torch.rand(C, device=dev); b = torch.rand(C, device=dev)
    y = ops.new_act(B,C,ho,ho,dev)
    res = {}
    for name, fn in (('u64', lambda: ops.gn_relu_upsample(x, st, g, b, (ho,ho), out=y)), ('u32', lambda: ops.gn_relu_upsample_sum([(x,st)], g, b, (ho,ho), out=y))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0,e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        res[name] = e0.elapsed_time(e1)/20
    gb = (B*C*hi*hi + B*C*ho*ho)*4/1e9
    print(C,hi,ho, {k: f'{v:.3f} ms {gb/v*1e3:.0f} GB/s' for k,v in res.items()})

# the semantic-branch tail: sum of three terms (+ fused 1x1 output layer) vs sum followed by lm_conv2d_nhwc_small
C_, h = 128, 288
xs = [ops.new_act(B, C_, s, s, dev).normal_() for s in (288, 144, 288)]
terms = [(x, ops.gn_stats(x)) for x in xs]
g, b = torch.rand(C_, device=dev), torch.rand(C_, device=dev)
for cout in (8, 1):
    w16 = ops.pack_small(torch.randn(cout, C_, 1, 1, device=dev))
    bias = torch.randn(cout, device=dev)
    res = {}
    for name, fn in (('sum then 1x1', lambda: ops.conv_small(ops.gn_relu_upsample_sum(terms, g, b, (h, h)), w16, cout, shift=bias)),
                     ('sum only', lambda: ops.gn_relu_upsample_sum(terms, g, b, (h, h))),
                     ('fused, sum not written', lambda: ops.gn_relu_upsample_sum(terms, g, b, (h, h), proj=(w16, bias, cout), keep_sum=False))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name] = round(e0.elapsed_time(e1) / 20, 3)
    print(f'128 -> {cout}:', res)
