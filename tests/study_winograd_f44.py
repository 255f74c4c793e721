#!/usr/bin/env python3
"""Numerics study for the next fp32 lever: Winograd F(4x4,3x3) - 36 multiplies per 16 outputs, 2.25 per output against the 4
of the F(2x2,3x3) the product runs - in exact fp32 arithmetic.  Its transforms carry constants up to 8 and down to 1/24, so its rounding
error is larger than F(2x2)'s (whose transforms only use 0, +-1, 1/2); the question is whether the parity contract survives: raw outputs
within 1e-4 of the tensor scale and every integer decision equal wherever the reference's own margin is >= 1e-4.

Emulated on the CPU in fp32 (torch), the way tests/study_split_precision.py priced the bf16x3 split before its kernel was written:
  Part 1: one layer (256 -> 256, 3x3, FPN-like input statistics), error vs fp64: fp32 direct, F(2x2,3x3), F(4x4,3x3) (U computed in fp64
          at pack time and rounded once, as a weight packer would).
  Part 2: the whole config-2 network on the G10 tile with EVERY 3x3 / stride-1 convolution with >= 64 input channels through F(4x4,3x3)
          (dilated layers as d x d interleaved plain convolutions, like the product): decision flips vs the reference golden.
Usage: python tests/study_winograd_f44.py > profiles/r3_f44_numerics_study.txt   (CPU only, ~ minutes)
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from lanemapping_amd import synth  # noqa: E402
from lanemapping_amd.boundary import build_net_from_config  # noqa: E402
from oracle import net_ref, decode_ref  # noqa: E402

torch.set_num_threads(8)
_conv = F.conv2d

# Lavin & Gray, "Fast Algorithms for Convolutional Neural Networks" (2015), F(4x4,3x3), interpolation points 0, +-1, +-2, inf
BT44 = torch.tensor([[4., 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0],
                     [0, 4, 0, -5, 0, 1]])
G44 = torch.tensor([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1.]],
                   dtype=torch.float64)
AT44 = torch.tensor([[1., 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]])
BT22 = torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
G22 = torch.tensor([[1., 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1.]], dtype=torch.float64)
AT22 = torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]])


def wino_plain(x, w, m, Bt, G, At):
    """F(m x m, 3x3), pad 1, stride 1, fp32 throughout except U = G g G^T (fp64, rounded once)."""
    B_, C, H, W = x.shape
    t = m + 2
    Hp, Wp = (H + m - 1) // m * m, (W + m - 1) // m * m
    xp = F.pad(x, (1, 1 + Wp - W, 1, 1 + Hp - H))
    d = xp.unfold(2, t, m).unfold(3, t, m)                                   # [B, C, ty, tx, t, t]
    V = torch.einsum('ij,bcyxjk,lk->bcyxil', Bt, d, Bt)
    U = torch.einsum('ij,ocjk,lk->ocil', G, w.double(), G).float()
    out = torch.empty((B_, w.shape[0], Hp // m, Wp // m, t, t))
    for i in range(t):                                                       # xi by xi: bounded memory
        for l in range(t):
            out[:, :, :, :, i, l] = torch.einsum('bcyx,oc->boyx', V[..., i, l], U[:, :, i, l])
    Y = torch.einsum('ij,boyxjk,lk->boyxil', At, out, At)
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(B_, w.shape[0], Hp, Wp)
    return y[:, :, :H, :W]


def conv_wino(x, w, b, dilation, m, Bt, G, At):
    d = dilation if isinstance(dilation, int) else dilation[0]
    if d == 1:
        y = wino_plain(x, w, m, Bt, G, At)
    else:                                                                    # d x d interleaved plain convolutions
        y = torch.empty((x.shape[0], w.shape[0], x.shape[2], x.shape[3]))
        for a in range(d):
            for c in range(d):
                y[:, :, a::d, c::d] = wino_plain(x[:, :, a::d, c::d], w, m, Bt, G, At)
    return y if b is None else y + b.view(1, -1, 1, 1)


def part1():
    g = torch.Generator().manual_seed(7)
    x = F.relu(torch.randn((1, 256, 64, 64), generator=g)) * 1.3
    w = torch.randn((256, 256, 3, 3), generator=g) / (256 * 9) ** 0.5
    ref = _conv(x.double(), w.double(), None, 1, 1).float()
    rows = []
    for name, y in (('fp32 direct', _conv(x, w, None, 1, 1)), ('fp32 Winograd F(2x2,3x3)', conv_wino(x, w, None, 1, 2, BT22, G22, AT22)),
                    ('fp32 Winograd F(4x4,3x3)', conv_wino(x, w, None, 1, 4, BT44, G44, AT44))):
        e = (y - ref).abs()
        rows.append(f'  {name:26s} max |err| {float(e.max()):.3e}  rms {float(e.pow(2).mean().sqrt()):.3e}  (output scale {float(ref.abs().max()):.2f})')
    return rows


def part2(m):
    net = build_net_from_config('Proj_polyline_fpn_vit_vertex_2', device='cpu')
    synth.fill_module_(net, 2021)
    sd = net.state_dict()
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g10_e2e.npz'))
    x = torch.from_numpy(synth.bev_batch([int(g['tile_seed'])], 1152))
    Bt, G, At = (BT44, G44, AT44) if m == 4 else (BT22, G22, AT22)
    n = [0]

    def patched(inp, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
        st = stride if isinstance(stride, int) else stride[0]
        dl = dilation if isinstance(dilation, int) else dilation[0]
        pd = padding if isinstance(padding, int) else padding[0]
        if weight.shape[2:] == (3, 3) and st == 1 and weight.shape[1] >= 64 and pd == dl and groups == 1:
            n[0] += 1
            return conv_wino(inp, weight, bias, dl, m, Bt, G, At)
        return _conv(inp, weight, bias, stride, padding, dilation, groups)
    F.conv2d = patched
    try:
        with torch.no_grad():
            raw = net_ref.detector_forward(sd, x)
    finally:
        F.conv2d = _conv
    d = decode_ref.decode_column_proposals({k: v.numpy() for k, v in raw.items()})
    rows = [f'  ({n[0]} convolutions replaced)']
    for k, gk in (('proposal_conf', 'proposal_conf'), ('ext2', 'ext2'), ('cls2', 'cls2'), ('offset2', 'offset2'), ('orient', 'orient_logits')):
        e = float(np.abs(raw[k].numpy() - g[gk]).max())
        sc = float(np.abs(g[gk]).max())
        rows.append(f'  {k:14s} max |err| vs reference {e:.3e}  (tensor scale {sc:.2f}; contract 1e-4 * max(1, scale) = {1e-4 * max(1.0, sc):.1e})')
    sem = d['semantic_seg'][0].numpy().astype(np.uint8).reshape(-1)
    bad = np.flatnonzero(sem != g['semantic_seg'][0].reshape(-1))
    low = set(g['sem_lowmargin'].tolist())
    rows.append(f'  semantic_seg   {bad.size} of {sem.size} pixels flip, {sum(1 for b in bad if int(b) not in low)} of them where the reference margin is >= 1e-4')
    ext = d['prop_v_ext'][0].numpy().astype(np.uint8).reshape(-1)
    bad = np.flatnonzero(ext != g['prop_v_ext'][0].reshape(-1))
    low = set(g['ext_lowmargin'].tolist())
    rows.append(f'  prop_v_ext     {bad.size} flips, {sum(1 for b in bad if int(b) not in low)} outside the margin')
    ci = raw['cls2'].argmax(-1)[0].numpy().reshape(-1)
    gi = g['cls2'].argmax(-1)[0].reshape(-1)
    bad = np.flatnonzero(ci != gi)
    rows.append(f'  column bin     {bad.size} flips, {int((g["cls2_margin"][0].reshape(-1)[bad] >= 1e-4).sum())} outside the margin')
    oi = raw['orient'].argmax(1)[0].numpy().reshape(-1)
    bad = np.flatnonzero(oi != g['orient'][0].reshape(-1))
    low = set(g['orient_lowmargin'].tolist())
    rows.append(f'  orient         {bad.size} flips, {sum(1 for b in bad if int(b) not in low)} outside the margin')
    return rows


if __name__ == '__main__':
    print('Part 1 - one layer, 256 -> 256 3x3, 64x64, error vs fp64:')
    print('\n'.join(part1()))
    for m in (2, 4):
        print(f'Part 2 - whole config-2 net on the G10 tile, every 3x3 / stride-1 / Cin >= 64 convolution through fp32 Winograd F({m}x{m},3x3) '
              f'(emulated), vs the reference golden:')
        print('\n'.join(part2(m)))
