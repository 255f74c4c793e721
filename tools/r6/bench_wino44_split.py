"""Per-layer timing of wino44_kernel: the exact fp32 kernel against the split-precision second line (fp16 x 2 terms, three products) on the
FPN's layer shapes, plus the error of both against an fp64 convolution on a crop.  Usage: python tools/r6/bench_wino44_split.py [B] [reps]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from lanemapping_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device('cuda:0')
SHAPES = [(256, 256, 288, 1), (256, 512, 144, 1), (256, 256, 144, 2), (256, 128, 288, 1), (128, 128, 144, 1), (128, 256, 144, 1), (64, 64, 288, 1)]


def timed(fn):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REPS):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / REPS


for cin, cout, hw, dil in SHAPES:
    g = torch.Generator().manual_seed(cin + cout + hw)
    x = ops.new_act(B, cin, hw, hw, dev).normal_()
    w = (torch.randn((cout, cin, 3, 3), generator=g) / (cin * 9) ** 0.5).to(dev)
    sc, sh = (torch.rand(cout, generator=g) + 0.5).to(dev), torch.randn(cout, generator=g).to(dev)
    res = ops.new_act(B, cout, hw, hw, dev).normal_()
    wu = ops.pack_wino44(w)
    wf, ws = ops.pack_wino44_fragments(wu), ops.pack_wino44_fragments_split(wu)
    ye, ys = ops.new_act(B, cout, hw, hw, dev), ops.new_act(B, cout, hw, hw, dev)
    te = timed(lambda: ops.conv_wino44(x, wf, cout, dil, scale=sc, shift=sh, res=res, act=ops.ACT_RELU, out=ye))
    ts = timed(lambda: ops.conv_wino44(x, ws, cout, dil, scale=sc, shift=sh, res=res, act=ops.ACT_RELU, out=ys))
    te2 = timed(lambda: ops.conv_wino44(x, wf, cout, dil, scale=sc, shift=sh, res=res, act=ops.ACT_RELU, out=ye))
    ts2 = timed(lambda: ops.conv_wino44(x, ws, cout, dil, scale=sc, shift=sh, res=res, act=ops.ACT_RELU, out=ys))
    c = 64
    xs = x[:1, :, :c + 2 * dil, :c + 2 * dil].double().cpu()
    want = F.conv2d(xs, w.double().cpu(), None, 1, 0, dil) * sc.double().cpu().view(1, -1, 1, 1) + sh.double().cpu().view(1, -1, 1, 1)
    want = F.relu(want + res[:1, :, dil:c + dil, dil:c + dil].double().cpu())
    ee = float((ye[:1, :, dil:c + dil, dil:c + dil].double().cpu() - want).abs().max())
    es = float((ys[:1, :, dil:c + dil, dil:c + dil].double().cpu() - want).abs().max())
    tiles = ops.lib().lm_winograd44_tiles(B, hw, hw, dil)
    ex = 2.0 * 36 * tiles * cin * cout
    print(f'{cin}->{cout} d{dil}@{hw} B{B}: exact {min(te, te2):.3f} ms ({ex / min(te, te2) / 1e9:.1f} TFLOP/s), split {min(ts, ts2):.3f} ms (x{min(te, te2) / min(ts, ts2):.2f}; '
          f'{3 * ex / min(ts, ts2) / 1e9:.0f} TFLOP/s of fp16 products executed = {3 * ex / min(ts, ts2) / 1e9 / 2500:.3f} of the 2.5 PFLOP/s fp16 peak) '
          f'err vs fp64 exact {ee:.1e} / split {es:.1e} (scale {float(want.abs().max()):.1f})', flush=True)
