import os, sys
import numpy as np
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from lanemapping_amd import ops
dev = torch.device('cuda:0')
B, cin, cout, H, W, dil = 1, 16, 32, 8, 48, 1
g = torch.Generator().manual_seed(5)
x = ops.new_act(B, cin, H, W, dev); x.copy_(torch.randn((B, cin, H, W), generator=g).to(dev))
w = (torch.randn((cout, cin, 3, 3), generator=g) / (cin * 9) ** 0.5).to(dev)
wu = ops.pack_wino44(w)
ws = ops.pack_wino44_fragments_split(wu)
sc = ops.split_scale(wu)
frag = ops.pack_wino44_fragments(wu * sc).cpu().numpy()
words = ws.words.cpu().numpy().view(np.uint32)
print('frag shape', frag.shape, 'scale', sc, 'max |U*sc|', np.abs(frag).max())
q = frag.reshape(-1, 4)[:4]
wq = words.reshape(-1, 4)[:4]
for i in range(4):
    hi01 = np.array([wq[i, 0] & 0xffff, wq[i, 0] >> 16], dtype=np.uint16).view(np.float16).astype(np.float32)
    hi23 = np.array([wq[i, 1] & 0xffff, wq[i, 1] >> 16], dtype=np.uint16).view(np.float16).astype(np.float32)
    lo01 = np.array([wq[i, 2] & 0xffff, wq[i, 2] >> 16], dtype=np.uint16).view(np.float16).astype(np.float32)
    lo23 = np.array([wq[i, 3] & 0xffff, wq[i, 3] >> 16], dtype=np.uint16).view(np.float16).astype(np.float32)
    print('quad', q[i], 'hi', hi01, hi23, 'lo', lo01, lo23, 'recon err', np.abs(np.concatenate([hi01 + lo01, hi23 + lo23]) - q[i]).max())
ref = F.conv2d(x.double(), w.double(), None, 1, dil, dil)
yf = ops.conv_wino44(x, ws, cout, dil)
yt = ops.conv_wino44_twin(x, wu, cout, dil, split=True)
print('ref', ref[0, :3, 2, :6].cpu().numpy())
print('fused', yf[0, :3, 2, :6].cpu().numpy())
print('twin', yt[0, :3, 2, :6].cpu().numpy())
r = (yf.double() / ref).flatten()
print('fused/ref quantiles', torch.quantile(r, torch.tensor([0.1, 0.5, 0.9], dtype=torch.float64, device=dev)).cpu().numpy())
