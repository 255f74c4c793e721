import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from lanemapping_amd import ops
dev = torch.device('cuda:0')
B, cin, cout, H, W, dil = 1, 128, 64, 53, 111, 1
for amp in (1.0, 2.6e-3):
    for use_scale in (False, True):
        g = torch.Generator().manual_seed(2001)
        x = ops.new_act(B, cin, H, W, dev); x.copy_((torch.randn((B, cin, H, W), generator=g) * amp).to(dev))
        w = (torch.randn((cout, cin, 3, 3), generator=g) / (cin * 9) ** 0.5).to(dev)
        sc = (torch.rand(cout, generator=g) + 0.5).to(dev) if use_scale else None
        sh = (torch.randn(cout, generator=g) * amp).to(dev)
        wu = ops.pack_wino44(w)
        ws = ops.pack_wino44_fragments_split(wu)
        yt = ops.conv_wino44_twin(x, wu, cout, dil, scale=sc, shift=sh, split=True)
        yf = ops.conv_wino44(x, ws, cout, dil, scale=sc, shift=sh)
        ref = F.conv2d(x.double(), w.double(), None, 1, dil, dil)
        if sc is not None: ref = ref * sc.double().view(1, -1, 1, 1)
        ref = ref + sh.double().view(1, -1, 1, 1)
        n_nan = int(torch.isnan(yt).sum())
        print(f'amp {amp} scale {use_scale}: twin NaNs {n_nan} of {yt.numel()}, fused err {float((yf.double()-ref).abs().max()):.2e}, twin err (finite) {float((yt.double()-ref).abs().nan_to_num(0).max()):.2e}, equal {torch.equal(yt, yf)}')
        if n_nan:
            idx = torch.nonzero(torch.isnan(yt))
            print('   NaN channels:', sorted(set(idx[:, 1].tolist()))[:20], 'rows', sorted(set(idx[:, 2].tolist()))[:10], 'cols', sorted(set(idx[:, 3].tolist()))[:10])
