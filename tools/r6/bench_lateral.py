#!/usr/bin/env python3
"""ms per launch of the FPN laterals at the headline batch (B = 16) under the current LM_CONV_LATERAL setting:
64 -> 256 @288^2 + bilinear(coarse 144^2) and 128 -> 256 @144^2 + residual.  A/B: run once with LM_CONV_LATERAL=1, once with =0."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lanemapping_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
g = torch.Generator().manual_seed(1)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


res = {'LM_CONV_LATERAL': os.environ.get('LM_CONV_LATERAL', '1'), 'B': B}
x = ops.new_act(B, 64, 288, 288, dev).normal_()
w = ops.pack_mfma((torch.randn(256, 64, 1, 1, generator=g) / 8).to(dev))
bias = torch.randn(256, generator=g).to(dev)
coarse = ops.new_act(B, 256, 144, 144, dev).normal_()
out = ops.new_act(B, 256, 288, 288, dev)
ms = timeit(lambda: ops.conv_mfma(x, w, 256, shift=bias, res_up=coarse, out=out))
gb = B * 288 * 288 * (64 + 256) * 4 / 1e9 + B * 144 * 144 * 256 * 4 / 1e9
res['64->256@288 + up'] = {'ms': ms, 'TBps_algorithmic': gb / ms, 'TFLOPs': 2 * B * 288 * 288 * 64 * 256 / ms / 1e9}
x = ops.new_act(B, 128, 144, 144, dev).normal_()
w = ops.pack_mfma((torch.randn(256, 128, 1, 1, generator=g) / 11).to(dev))
r = ops.new_act(B, 256, 144, 144, dev).normal_()
out = ops.new_act(B, 256, 144, 144, dev)
ms = timeit(lambda: ops.conv_mfma(x, w, 256, shift=bias, res=r, out=out))
gb = B * 144 * 144 * (128 + 256 + 256) * 4 / 1e9
res['128->256@144 + res'] = {'ms': ms, 'TBps_algorithmic': gb / ms, 'TFLOPs': 2 * B * 144 * 144 * 128 * 256 / ms / 1e9}
print(json.dumps(res))
