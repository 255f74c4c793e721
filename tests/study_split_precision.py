#!/usr/bin/env python3
"""Numerics study SURVEY §7 / VERDICT r1 item 5 ask for ("BF16x3 split only if parity holds"): fp32 convolution through split
low-precision MFMA operands, emulated on the CPU (products of bf16 / fp16 pieces are exact in fp32; accumulation in fp32 like the
matrix cores' accumulators).

  bf16x3 / 6 products : a = a1 + a2 + a3 (bf16 each), a*b ~ a1b1 + a1b2 + a2b1 + a1b3 + a2b2 + a3b1          -> 6 bf16 MFMAs per fp32 one
  fp16x2 / 3 products : a = a1 + a2 (fp16 each, tensors pre-scaled by a power of two), a*b ~ a1b1 + a1b2 + a2b1 -> 3 fp16 MFMAs

Part 1: one layer (256 -> 256, 3x3, input statistics of an FPN activation), error vs fp64 next to plain fp32 (direct) and fp32
        Winograd F(2x2,3x3).
Part 2: the whole config-2 network on the G10 tile with EVERY 3x3 / stride-1 / Cin >= 128 convolution replaced by the split form:
        decision flips vs the reference golden (same accounting as the GPU parity test: flips only count where the reference's own
        margin is >= 1e-4).
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


def split_bf16(x, n=3):
    out, r = [], x
    for _ in range(n):
        p = r.to(torch.bfloat16).to(torch.float32)
        out.append(p)
        r = r - p
    return out


def split_fp16(x, n=2):
    s = 2.0 ** (14 - int(torch.ceil(torch.log2(x.abs().max().clamp_min(1e-30)))))       # top of the fp16 range
    out, r = [], x * s
    for _ in range(n):
        p = r.to(torch.float16).to(torch.float32)
        out.append(p)
        r = r - p
    return out, s


def conv_bf16x3(x, w, b=None, stride=1, padding=0, dilation=1):
    xs, ws = split_bf16(x), split_bf16(w)
    y = None
    for i, j in ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)):          # small terms first
        t = _conv(xs[i], ws[j], None, stride, padding, dilation)
        y = t if y is None else y + t
    return y if b is None else y + b.view(1, -1, 1, 1)


def conv_fp16x2(x, w, b=None, stride=1, padding=0, dilation=1):
    (x1, x2), sx = split_fp16(x)
    (w1, w2), sw = split_fp16(w)
    y = (_conv(x2, w1, None, stride, padding, dilation) + _conv(x1, w2, None, stride, padding, dilation)) + _conv(x1, w1, None, stride, padding, dilation)
    y = y / (sx * sw)
    return y if b is None else y + b.view(1, -1, 1, 1)


def wino_f32(x, w, b=None, padding=1):
    """F(2x2,3x3) in fp32 with the transform matrices of csrc/conv_wino.hip (for the error column only)."""
    B_, C, H, W = x.shape
    G = torch.tensor([[1., 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1.]])
    Bt = torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
    At = torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]])
    U = torch.einsum('ij,ocjk,lk->oicl', G, w, G)                                      # [O, 4, C, 4]
    xp = F.pad(x, (1, 1 + W % 2, 1, 1 + H % 2))
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                              # [B, C, Ty, Tx, 4, 4]
    V = torch.einsum('ij,bctxjk,lk->bctxil', Bt, d, Bt)
    M = torch.einsum('bctxil,oicl->botxil', V, U)
    Y = torch.einsum('ij,botxjk,lk->botxil', At, M, At)                                 # [B, O, Ty, Tx, 2, 2]
    y = Y.permute(0, 1, 2, 4, 3, 5).reshape(B_, w.shape[0], Y.shape[2] * 2, Y.shape[3] * 2)[:, :, :H, :W]
    return y if b is None else y + b.view(1, -1, 1, 1)


def part1():
    g = torch.Generator().manual_seed(7)
    x = F.relu(torch.randn((1, 256, 64, 64), generator=g)) * 1.3
    w = torch.randn((256, 256, 3, 3), generator=g) / (256 * 9) ** 0.5
    ref = _conv(x.double(), w.double(), None, 1, 1).float()
    scale = float(ref.abs().max())
    rows = []
    for name, y in (('fp32 direct (MKL-DNN order)', _conv(x, w, None, 1, 1)), ('fp32 Winograd F(2x2,3x3)', wino_f32(x, w)),
                    ('bf16x3, 6 products, direct', conv_bf16x3(x, w, None, 1, 1)), ('fp16x2, 3 products, direct', conv_fp16x2(x, w, None, 1, 1))):
        e = (y - ref).abs()
        rows.append(f'  {name:32s} max |err| {float(e.max()):.3e}  rms {float(e.pow(2).mean().sqrt()):.3e}   (output scale {scale:.2f})')
    return rows


def part2(kind):
    net = build_net_from_config('Proj_polyline_fpn_vit_vertex_2', device='cpu')
    synth.fill_module_(net, 2021)
    sd = net.state_dict()
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'g10_e2e.npz'))
    x = torch.from_numpy(synth.bev_batch([int(g['tile_seed'])], 1152))
    fn = conv_bf16x3 if kind == 'bf16x3' else conv_fp16x2

    def patched(inp, weight, bias=None, stride=1, padding=0, dilation=1, groups=1):
        st = stride if isinstance(stride, int) else stride[0]
        if weight.shape[2:] == (3, 3) and st == 1 and weight.shape[1] >= 128:
            return fn(inp, weight, bias, stride, padding, dilation)
        return _conv(inp, weight, bias, stride, padding, dilation, groups)
    F.conv2d = patched
    try:
        with torch.no_grad():
            raw = net_ref.detector_forward(sd, x)
    finally:
        F.conv2d = _conv
    d = decode_ref.decode_column_proposals({k: v.numpy() for k, v in raw.items()})
    rows = []
    for k, gk in (('proposal_conf', 'proposal_conf'), ('ext2', 'ext2'), ('cls2', 'cls2'), ('offset2', 'offset2'), ('orient', 'orient_logits')):
        e = float(np.abs(raw[k].numpy() - g[gk]).max())
        rows.append(f'  {k:14s} max |err| vs reference {e:.3e}  (tensor scale {float(np.abs(g[gk]).max()):.2f})')
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


def part3():
    """Winograd-domain split: V = B^T d B and U = G g G^T (fp32) split into 3 bf16 pieces each (truncation split = what the kernel's
    transform phase does with bit masks; every piece exact), M = sum over 6 piece products, fp32 accumulation; vs fp64 and vs the
    fp32 Winograd."""
    g = torch.Generator().manual_seed(7)
    x = F.relu(torch.randn((1, 256, 64, 64), generator=g)) * 1.3
    w = torch.randn((256, 256, 3, 3), generator=g) / (256 * 9) ** 0.5
    ref = _conv(x.double(), w.double(), None, 1, 1).float()
    G = torch.tensor([[1., 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1.]])
    Bt = torch.tensor([[1., 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]])
    At = torch.tensor([[1., 1, 1, 0], [0, 1, -1, -1]])
    U = torch.einsum('ij,ocjk,lk->oicl', G, w, G)
    d = F.pad(x, (1, 1, 1, 1)).unfold(2, 4, 2).unfold(3, 4, 2)
    V = torch.einsum('ij,bctxjk,lk->bctxil', Bt, d, Bt)

    def trunc_split(t, n=3):
        out, r = [], t
        for _ in range(n):
            p = (r.view(torch.int32) & -65536).view(torch.float32)          # keep sign, exponent, top 7 mantissa bits = a bf16 value
            out.append(p)
            r = r - p
        return out
    rows = []
    for name, split in (('RNE split', split_bf16), ('truncation split', trunc_split)):
        Vs, Us = split(V), split(U)
        M = None
        for i, j in ((2, 0), (1, 1), (0, 2), (1, 0), (0, 1), (0, 0)):
            t = torch.einsum('bctxil,oicl->botxil', Vs[i], Us[j])
            M = t if M is None else M + t
        Y = torch.einsum('ij,botxjk,lk->botxil', At, M, At)
        y = Y.permute(0, 1, 2, 4, 3, 5).reshape(1, 256, 64, 64)
        e = (y - ref).abs()
        rows.append(f'  Winograd F(2x2,3x3), bf16x3 ({name}), 6 products  max |err| {float(e.max()):.3e}  rms {float(e.pow(2).mean().sqrt()):.3e}')
        assert all(torch.equal(sum(Vs), V) for _ in (0,)), 'the three pieces must reproduce V exactly'
    return rows


if __name__ == '__main__':
    print('Part 1 - one layer, 256 -> 256 3x3, 64x64, error vs fp64:')
    print('\n'.join(part1()))
    print('Part 3 - the same layer through Winograd with the transformed operands split (what a bf16x3 GEMM kernel executes):')
    print('\n'.join(part3()))
    for kind in ('bf16x3', 'fp16x2'):
        print(f'Part 2 - whole config-2 net on the G10 tile, wide 3x3 convolutions through {kind} (emulated), vs the reference golden:')
        print('\n'.join(part2(kind)))
