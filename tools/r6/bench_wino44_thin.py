"""Per-layer timing of the thin layers: wino44_kernel against wino44t_kernel (16-tile workgroups, two per CU), B = 16 unless given."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from lanemapping_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
REPS = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda:0')


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


for cin, cout, hw, dil, use_res in [(64, 64, 288, 1, False), (64, 64, 288, 1, True), (128, 128, 144, 1, False), (128, 128, 144, 1, True), (128, 256, 144, 1, False)]:
    g = torch.Generator().manual_seed(cin + cout + hw)
    x = ops.new_act(B, cin, hw, hw, dev).normal_()
    w = (torch.randn((cout, cin, 3, 3), generator=g) / (cin * 9) ** 0.5).to(dev)
    sc, sh = (torch.rand(cout, generator=g) + 0.5).to(dev), torch.randn(cout, generator=g).to(dev)
    res = ops.new_act(B, cout, hw, hw, dev).normal_() if use_res else None
    wu = ops.pack_wino44(w)
    wf, wt = ops.pack_wino44_fragments(wu), ops.pack_wino44_fragments16(wu)
    ye, yt = ops.new_act(B, cout, hw, hw, dev), ops.new_act(B, cout, hw, hw, dev)
    te = min(timed(lambda: ops.conv_wino44(x, wf, cout, dil, scale=sc, shift=sh, res=res, act=ops.ACT_RELU, out=ye)) for _ in range(2))
    tt = min(timed(lambda: ops.conv_wino44(x, wt, cout, dil, scale=sc, shift=sh, res=res, act=ops.ACT_RELU, out=yt)) for _ in range(2))
    tiles = ops.lib().lm_winograd44_tiles(B, hw, hw, dil)
    ex = 2.0 * 36 * tiles * cin * cout
    print(f'{cin}->{cout} d{dil}@{hw} B{B} res={use_res}: wino44_kernel {te:.3f} ms ({ex / te / 1e9:.1f} TFLOP/s = {ex / te / 1e9 / 157.3:.2f}), wino44t_kernel {tt:.3f} ms '
          f'({ex / tt / 1e9:.1f} TFLOP/s = {ex / tt / 1e9 / 157.3:.2f}; x{te / tt:.2f}), max |difference| {float((ye - yt).abs().max()):.1e}', flush=True)
