"""Per-layer timing of the Winograd F(4x4,3x3) kernel (csrc/conv_wino44.hip) against the direct implicit-GEMM kernel (csrc/conv_mfma.hip) on the
FPN's layer shapes, plus error of both against an fp64 convolution on a crop.  Usage: python tools/r4/bench_wino44.py [B] [reps]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from lanemapping_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 10
dev = torch.device('cuda:0')
SHAPES = [(256, 256, 288, 1), (256, 512, 144, 1), (256, 256, 144, 2), (128, 128, 144, 1), (128, 256, 144, 1), (64, 64, 288, 1)]
if os.environ.get('SHAPES'):
    SHAPES = [tuple(int(v) for v in s.split(',')) for s in os.environ['SHAPES'].split(';')]


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


out = []
for cin, cout, hw, dil in SHAPES:
    g = torch.Generator().manual_seed(cin + cout + hw)
    x = ops.new_act(B, cin, hw, hw, dev).normal_()
    w = (torch.randn((cout, cin, 3, 3), generator=g) / (cin * 9) ** 0.5).to(dev)
    sc, sh = (torch.rand(cout, generator=g) + 0.5).to(dev), torch.randn(cout, generator=g).to(dev)
    res = ops.new_act(B, cout, hw, hw, dev).normal_()
    wp = ops.pack_mfma(w)
    wf4 = ops.pack_wino44_fragments(ops.pack_wino44(w))
    y2 = ops.new_act(B, cout, hw, hw, dev)
    y4 = ops.new_act(B, cout, hw, hw, dev)
    t2 = timed(lambda: ops.conv_mfma(x, wp, cout, 3, 3, 1, dil, dil, scale=sc, shift=sh, res=res, act=ops.ACT_RELU, out=y2))
    t4 = timed(lambda: ops.conv_wino44(x, wf4, cout, dil, scale=sc, shift=sh, res=res, act=ops.ACT_RELU, out=y4))
    # error against fp64 on one image crop (the whole tensor would take the CPU minutes)
    c = 64
    xs = x[:1, :, :c + 2 * dil, :c + 2 * dil].double().cpu()
    want = F.conv2d(xs, w.double().cpu(), None, 1, 0, dil) * sc.double().cpu().view(1, -1, 1, 1) + sh.double().cpu().view(1, -1, 1, 1)
    want = F.relu(want + res[:1, :, dil:c + dil, dil:c + dil].double().cpu())
    e2 = float((y2[:1, :, dil:c + dil, dil:c + dil].double().cpu() - want).abs().max())
    e4 = float((y4[:1, :, dil:c + dil, dil:c + dil].double().cpu() - want).abs().max())
    flops = 2.0 * B * hw * hw * cin * cout * 9
    tiles = ops.lib().lm_winograd44_tiles(B, hw, hw, dil)
    ex = 2.0 * 36 * tiles * cin * cout
    out.append(f'{cin}->{cout} d{dil}@{hw} B{B}: direct {t2:.3f} ms, F(4x4) {t4:.3f} ms, (x{t2 / t4:.2f}; executed {ex / t4 / 1e9:.1f} TFLOP/s = {ex / t4 / 1e9 / 157.3:.2f} of peak, '
               f'direct-equivalent {flops / t4 / 1e9:.0f}) err vs fp64 {e2:.1e} / {e4:.1e} (scale {float(want.abs().max()):.1f})')
    print(out[-1], flush=True)
