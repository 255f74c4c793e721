#!/usr/bin/env python3
"""Micro-bench of the non-Winograd kernels of a step that round 4 touched: the one-term GN + ReLU + up-sampling call (s4), the tiny-K
1x1 convolutions with the `_upsample_add` residual, stem + max-pool.  usage: bench_tail.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from lanemapping_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8


def timed(fn, n=20):
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


# s4: 256 channels, 144^2 -> 288^2
x = ops.new_act(B, 256, 144, 144, dev).normal_()
st = ops.gn_stats(x)
g, bt = torch.rand(256, device=dev), torch.rand(256, device=dev)
y = ops.new_act(B, 256, 288, 288, dev)
gb = (x.numel() + y.numel()) * 4 / 1e9
for name, fn in (('per-output kernel (lm_gn_relu_upsample)', lambda: ops.gn_relu_upsample(x, st, g, bt, (288, 288), out=y)),
                 ('one-term call (lm_gn_relu_upsample_sum)', lambda: ops.gn_relu_upsample_sum([(x, st)], g, bt, (288, 288), out=y))):
    ms = timed(fn)
    print(f'gn+relu+up 256ch 144->288 B{B}: {name}: {ms:.3f} ms  {gb / ms * 1e3:.0f} GB/s')

# tiny-K 1x1 convolutions
for cin, cout, hw, up in ((64, 256, 288, True), (128, 256, 144, True), (256, 256, 144, False), (64, 256, 288, False)):
    xi = ops.new_act(B, cin, hw, hw, dev).normal_()
    w = ops.pack_mfma(torch.randn(cout, cin, 1, 1, device=dev) / cin ** 0.5)
    bias = torch.randn(cout, device=dev)
    out = ops.new_act(B, cout, hw, hw, dev)
    coarse = ops.new_act(B, cout, hw // 2, hw // 2, dev).normal_() if up else None
    ms = timed(lambda: ops.conv_mfma(xi, w, cout, shift=bias, res_up=coarse, out=out))
    gb = (xi.numel() + out.numel() + (coarse.numel() if up else 0)) * 4 / 1e9
    print(f'conv1x1 {cin}->{cout} @{hw} B{B} {"+ upsample_add" if up else ""}: {ms:.3f} ms  {gb / ms * 1e3:.0f} GB/s  '
          f'{2.0 * B * hw * hw * cin * cout / ms / 1e9:.1f} TFLOP/s')

# stem + max-pool on a u8 tile
t = torch.randint(0, 255, (B, 1152, 1152, 3), dtype=torch.uint8, device=dev)
w7 = torch.randn(7, 7, 3, 64, device=dev) / 12
s, sh = torch.rand(64, device=dev) + 0.5, torch.randn(64, device=dev)
ms_s = timed(lambda: ops.stem(t, w7, s, sh))
c1 = ops.stem(t, w7, s, sh)
ms_p = timed(lambda: ops.maxpool3x3s2(c1))
print(f'stem (u8) B{B}: {ms_s:.3f} ms   max-pool: {ms_p:.3f} ms')

# endpoint top-K (radix select over the cropped 1152^2 logit map)
from lanemapping_amd import decode  # noqa: E402
lg = torch.randn(B, 1, 1152, 1152, device=dev)
ms = timed(lambda: ops.endp_topk(lg, decode.TOPK, decode.CLIP))
print(f'endp_topk B{B}: {ms:.3f} ms')

# head: proposal tokens (8 x 8 bilinear pooling of the segmentation map x the 16-channel row features)
seg = torch.randn(B, 1, 288, 288, device=dev)
rowf = ops.new_act(B, 16, 144, 144, dev).normal_()
ms = timed(lambda: ops.head_tokens(seg, rowf, 72, 2, 4, -0.3))
print(f'head_tokens B{B}: {ms:.3f} ms')

# head convolutions: 16 -> 16 3x3 @288^2 stride 1 / 2, 16 -> 8 @144^2
for hw, cout, stride in ((288, 16, 1), (288, 16, 2), (144, 8, 1)):
    xi = ops.new_act(B, 16, hw, hw, dev).normal_()
    w16 = ops.pack_small(torch.randn(cout, 16, 3, 3, device=dev) / 12)
    bias = torch.randn(cout, device=dev)
    ms = timed(lambda: ops.conv_small(xi, w16, cout, 3, 3, stride, 1, shift=bias))
    print(f'small conv 16->{cout} 3x3 s{stride} @{hw} B{B}: {ms:.3f} ms')

# head: second Conv1d of ext2 / cls2 / offset2 on the hidden rows (M = B * 72 * 144 rows of 3 x 100 floats)
hid = torch.randn(B * 72 * 144, 300, device=dev)
w2, b2 = torch.randn(23, 100, device=dev) / 10, torch.randn(23, device=dev)
ms = timed(lambda: ops.head_stage2(hid, 100, w2, b2, B, 72, 144))
print(f'head_stage2 B{B}: {ms:.3f} ms')
