"""ms per launch of lm_conv3x3_winograd44_f32 (wino44_kernel) on the FPN's layer shapes with residual + ReLU; usage: w44_time.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from lanemapping_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device('cuda:0')
for cin, cout, hw, dil in [(256, 256, 288, 1), (256, 512, 144, 1), (256, 256, 144, 2), (256, 128, 288, 1), (128, 128, 144, 1), (128, 256, 144, 1), (64, 64, 288, 1)]:
    x = ops.new_act(B, cin, hw, hw, dev).normal_()
    w = torch.randn((cout, cin, 3, 3), device=dev) / (cin * 9) ** 0.5
    wf = ops.pack_wino44_fragments(ops.pack_wino44(w))
    res = ops.new_act(B, cout, hw, hw, dev).normal_()
    y = ops.new_act(B, cout, hw, hw, dev)
    for _ in range(2):
        ops.conv_wino44(x, wf, cout, dil, res=res, act=ops.ACT_RELU, out=y)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        ops.conv_wino44(x, wf, cout, dil, res=res, act=ops.ACT_RELU, out=y)
    b.record()
    torch.cuda.synchronize()
    tiles = ops.lib().lm_winograd44_tiles(B, hw, hw, dil)
    ms = a.elapsed_time(b) / 10
    print(f'{cin}->{cout} d{dil}@{hw} B{B}: {ms:.3f} ms  ({2.0 * 36 * tiles * cin * cout / ms / 1e9 / 157.3:.3f} of peak)', flush=True)
