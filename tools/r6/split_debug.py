import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from lanemapping_amd import ops
dev = torch.device('cuda:0')
for (B, cin, cout, H, W, dil) in [(1, 16, 32, 8, 48, 1), (1, 16, 64, 8, 48, 1), (1, 32, 64, 8, 48, 1), (1, 128, 64, 53, 111, 1)]:
    g = torch.Generator().manual_seed(5)
    x = ops.new_act(B, cin, H, W, dev); x.copy_(torch.randn((B, cin, H, W), generator=g).to(dev))
    w = (torch.randn((cout, cin, 3, 3), generator=g) / (cin * 9) ** 0.5).to(dev)
    wu = ops.pack_wino44(w)
    ws = ops.pack_wino44_fragments_split(wu)
    ref = F.conv2d(x.double(), w.double(), None, 1, dil, dil)
    yt = ops.conv_wino44_twin(x, wu, cout, dil, split=True)
    yf = ops.conv_wino44(x, ws, cout, dil)
    ye = ops.conv_wino44(x, ops.pack_wino44_fragments(wu), cout, dil)
    print(f'{cin}->{cout} {H}x{W}: twin-split err {float((yt.double()-ref).abs().max()):.3e}  fused-split err {float((yf.double()-ref).abs().max()):.3e}  exact err {float((ye.double()-ref).abs().max()):.3e}  post {ws.post}')
    d = (yf.double() - ref).abs()
    # error per output channel block / per tile column to see the structure
    print('   fused-split err per 8-channel group:', [f'{float(d[0, c:c+8].max()):.1e}' for c in range(0, min(cout, 64), 8)])
    print('   fused-split err per 4-col group (first 12):', [f'{float(d[0, :, :, c:c+4].max()):.1e}' for c in range(0, 48, 4)])
    # one-hot input channel test: which input channels contribute correctly?
    if cin == 16:
        for ch in range(16):
            x1 = torch.zeros_like(x); x1[:, ch] = x[:, ch]
            r1 = F.conv2d(x1.double(), w.double(), None, 1, dil, dil)
            y1 = ops.conv_wino44(x1, ws, cout, dil)
            print(f'      only input channel {ch}: err {float((y1.double()-r1).abs().max()):.2e} (scale {float(r1.abs().max()):.2f})')
