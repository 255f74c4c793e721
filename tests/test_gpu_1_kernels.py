"""GPU parity tests (-m gpu), file 1 of 4: single kernels and C-ABI entry points against a torch fp32 / fp64 reference or the
oracle.  Tolerances: fp32 features within 1e-4 of the tensor scale (different but fixed summation order), every integer / index
output bit-exact.  The four files run in name order under `pytest -x`: kernels, goldens, configs / Runner, then every test that
launches `bench.py` as a subprocess (test_gpu_9_bench.py) - a harness assertion there can no longer hide a parity test."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cases
from gpu_common import ROOT, _close, _lidar_module, _rowref_head
from lanemapping_amd import synth

pytestmark = pytest.mark.gpu


# ----------------------------------------------------------------------------------------------- kernels
@pytest.mark.parametrize('cin,cout,k,stride,dil,hw,bn,res,relu', [
    (64, 64, 3, 1, 1, (40, 56), True, True, True),
    (64, 128, 3, 2, 1, (40, 56), True, False, True),
    (256, 256, 3, 1, 2, (24, 24), True, True, True),
    (256, 64, 1, 1, 1, (24, 24), False, False, False),
    (128, 256, 1, 1, 1, (17, 23), False, True, False),      # ragged M (not a multiple of the 128-row tile)
    (64, 128, 1, 2, 1, (40, 56), True, False, False),
    (256, 128, 3, 1, 1, (20, 36), False, False, False),
])
def test_conv_mfma_vs_torch(dev, cin, cout, k, stride, dil, hw, bn, res, relu):
    from lanemapping_amd import ops
    B = 2
    g = torch.Generator().manual_seed(cin * 7 + cout + k)
    x = torch.randn(B, cin, *hw, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    pad = dil * (k // 2)
    ref = F.conv2d(x, w, None, stride, pad, dil)
    scale = shift = None
    if bn:
        scale = torch.rand(cout, generator=g) + 0.5
        shift = torch.randn(cout, generator=g)
        ref = ref * scale.view(1, -1, 1, 1) + shift.view(1, -1, 1, 1)
    r = None
    if res:
        r = torch.randn(ref.shape, generator=g)
        ref = ref + r
    if relu:
        ref = F.relu(ref)
    y = ops.conv_mfma(x.to(dev), ops.pack_mfma(w.to(dev)), cout, k, k, stride, pad, dil,
                      scale=None if scale is None else scale.to(dev), shift=None if shift is None else shift.to(dev),
                      res=None if r is None else r.to(dev), act=ops.ACT_RELU if relu else ops.ACT_NONE)
    _close(y, ref, 1e-5, 'conv_mfma')
    y2 = ops.conv_mfma(x.to(dev), ops.pack_mfma(w.to(dev)), cout, k, k, stride, pad, dil,
                       scale=None if scale is None else scale.to(dev), shift=None if shift is None else shift.to(dev),
                       res=None if r is None else r.to(dev), act=ops.ACT_RELU if relu else ops.ACT_NONE)
    assert torch.equal(y, y2), 'conv_mfma must be deterministic'


def test_linear_mfma_gelu_bias_res(dev):
    from lanemapping_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(650, 512, generator=g)
    w = torch.randn(300, 512, generator=g) / 512 ** 0.5
    b = torch.randn(300, generator=g)
    ref = F.gelu(F.linear(x, w, b))
    out = torch.zeros(650, 320, device=dev)
    ops.linear_mfma(x.to(dev), ops.pack_mfma(w.to(dev)), 300, shift=b.to(dev), act=ops.ACT_GELU, out=out)
    _close(out[:, :300], ref, 1e-5, 'linear+gelu')
    assert float(out[:, 300:].abs().max()) == 0.0, 'columns beyond n_out must stay untouched'


def test_small_conv_and_stem(dev, synth_sd):
    from lanemapping_amd import ops
    g = torch.Generator().manual_seed(9)
    x = torch.randn(2, 16, 30, 44, generator=g)
    w = torch.randn(11, 16, 3, 3, generator=g) / 12
    b = torch.randn(11, generator=g)
    ref = F.conv2d(F.relu(x), w, b, 2, 1)
    y = ops.conv_small(x.to(dev), ops.pack_small(w.to(dev)), 11, 3, 3, 2, 1, shift=b.to(dev), pre_relu=True)
    _close(y, ref, 1e-5, 'small conv')
    # stem + max-pool against the oracle's first stage
    xs = torch.from_numpy(synth.bev_batch([7], 96))
    p = 'pcencoder.fpn'
    c1 = F.relu(F.batch_norm(F.conv2d(xs, synth_sd[p + '.conv1.weight'], None, 2, 3), synth_sd[p + '.bn1.running_mean'],
                             synth_sd[p + '.bn1.running_var'], synth_sd[p + '.bn1.weight'], synth_sd[p + '.bn1.bias'], False, 0., 1e-5))
    ref = F.max_pool2d(c1, 3, 2, 1)
    bn = torch.nn.BatchNorm2d(64)
    bn.load_state_dict({k: synth_sd[f'{p}.bn1.{k}'] for k in ('weight', 'bias', 'running_mean', 'running_var', 'num_batches_tracked')})
    s, sh = ops.fold_bn(bn)
    y = ops.maxpool3x3s2(ops.stem(xs.to(dev), synth_sd[p + '.conv1.weight'].permute(2, 3, 1, 0).contiguous().to(dev), s.to(dev), sh.to(dev)))
    _close(y, ref, 1e-5, 'stem+pool')


def test_gn_relu_upsample(dev):
    from lanemapping_amd import ops
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 128, 20, 28, generator=g) * 3 + 1
    gamma, beta = torch.rand(128, generator=g) + 0.5, torch.randn(128, generator=g)
    ref = F.interpolate(F.relu(F.group_norm(x, 128, gamma, beta, 1e-5)), size=(40, 56), mode='bilinear', align_corners=True)
    xd = x.to(dev)
    st = ops.gn_stats(xd)
    y = ops.gn_relu_upsample(xd, st, gamma.to(dev), beta.to(dev), (40, 56))
    _close(y, ref, 1e-5, 'gn+relu+up')
    y = ops.gn_relu_upsample(xd, st, gamma.to(dev), beta.to(dev), (40, 56), out=y, accumulate=True)
    _close(y, 2 * ref, 1e-5, 'accumulate')
    for size in ((80, 112), (81, 113), (37, 4)):              # 16-byte vector path (Wo % 4 == 0) and the scalar one
        up = ops.upsample_to_chw(xd[:, :3], size)
        _close(up, F.interpolate(x[:, :3], size=size, mode='bilinear', align_corners=True), 1e-5, f'to_chw {size}')


def test_gn_relu_upsample_sum(dev):
    """`s2 + s3 + s4` of a semantic branch in one pass: equals the torch expression and, bit for bit, three accumulating calls."""
    from lanemapping_amd import ops
    g = torch.Generator().manual_seed(12)
    shapes = [(40, 56), (20, 28), (40, 56)]
    xs = [torch.randn(2, 64, h, w, generator=g) * (k + 1) + 0.5 * k for k, (h, w) in enumerate(shapes)]
    gamma, beta = torch.rand(64, generator=g) + 0.5, torch.randn(64, generator=g)
    ref = sum(F.interpolate(F.relu(F.group_norm(x, 64, gamma, beta, 1e-5)), size=(40, 56), mode='bilinear', align_corners=True) for x in xs)
    xd = [x.to(dev) for x in xs]
    terms = [(x, ops.gn_stats(x)) for x in xd]
    gd, bd = gamma.to(dev), beta.to(dev)
    y = ops.gn_relu_upsample_sum(terms, gd, bd, (40, 56))
    _close(y, ref, 1e-5, 'sum of 3 terms')
    z = ops.gn_relu_upsample(terms[0][0], terms[0][1], gd, bd, (40, 56))
    for x, st in terms[1:]:
        z = ops.gn_relu_upsample(x, st, gd, bd, (40, 56), out=z, accumulate=True)
    assert torch.equal(y, z), float((y - z).abs().max())
    # fused 1x1 output layer (feature_layer 64 -> 8 here, and a single-channel one) with and without writing the sum
    for cout in (8, 1, 5):
        w = torch.randn(cout, 64, 1, 1, generator=g) / 8
        b = torch.randn(cout, generator=g)
        want = F.conv2d(ref, w, b)
        w16 = ops.pack_small(w.to(dev))
        ysum, y1 = ops.gn_relu_upsample_sum(terms, gd, bd, (40, 56), proj=(w16, b.to(dev), cout))
        assert torch.equal(ysum, y)
        _close(y1, want, 1e-5, f'fused 1x1 ({cout})')
        only = ops.gn_relu_upsample_sum(terms, gd, bd, (40, 56), proj=(w16, b.to(dev), cout), keep_sum=False)
        assert torch.equal(only, y1)
        _close(ops.conv_small(y, w16, cout, shift=b.to(dev)), want, 1e-5, 'unfused 1x1')
    y2 = ops.gn_relu_upsample_sum(terms[:2], gd, bd, (40, 56))
    _close(y2, ref - F.interpolate(F.relu(F.group_norm(xs[2], 64, gamma, beta, 1e-5)), size=(40, 56), mode='bilinear', align_corners=True),
           1e-5, 'sum of 2 terms')


@pytest.mark.parametrize('B,C,hi,wi,ho,wo,ld', [(2, 256, 36, 36, 72, 72, 256), (1, 128, 37, 21, 75, 50, 128), (2, 64, 20, 28, 56, 57, 96),
                                                 (1, 256, 144, 144, 288, 288, 384), (2, 32, 9, 5, 40, 56, 32)])
def test_gn_relu_upsample_one_term_lds_block(dev, B, C, hi, wi, ho, wo, ld):
    """The one-term up-sampling call (gn_relu_up_lds_kernel: source block normalised once and staged in LDS) against the per-output
    kernel of lm_gn_relu_upsample, bit for bit: ragged tiles (Ho % 8, Wo % 16 != 0), channel slices of a wider tensor (ld > C), scales
    below 1/2, and the s4 shape of the semantic branches."""
    from lanemapping_amd import ops
    g = torch.Generator().manual_seed(B * 1000 + C + hi)
    wide = (torch.randn(B, hi, wi, ld, generator=g) * 2 + 0.3).to(dev)
    x = wide[..., ld - C:].permute(0, 3, 1, 2)                    # NHWC-stored channel slice, [B,C,H,W] view
    dense = x.contiguous(memory_format=torch.channels_last)
    st = ops.gn_stats(dense)
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), torch.randn(C, generator=g).to(dev)
    want = ops.gn_relu_upsample(dense, st, gamma, beta, (ho, wo))
    got = ops.gn_relu_upsample_sum([(x, st)], gamma, beta, (ho, wo))
    assert torch.equal(got, want), float((got - want).abs().max())
    ref = F.interpolate(F.relu(F.group_norm(dense.float(), C, gamma, beta, 1e-5)), size=(ho, wo), mode='bilinear', align_corners=True)
    _close(got, ref, 1e-5, 'one term, LDS block')


@pytest.mark.parametrize('cin,cout,B,h,w,mode', [(64, 256, 2, 72, 72, 'up'), (128, 256, 1, 36, 40, 'up'), (64, 128, 2, 40, 24, 'res'),
                                                 (64, 200, 1, 33, 17, 'up'), (256, 256, 1, 36, 36, 'up'), (64, 256, 1, 24, 24, 'rows')])
def test_conv1x1_lateral_residuals(dev, cin, cout, B, h, w, mode):
    """1x1 convolutions of the FPN's lateral layers (tiny-K 64 x 64 tiles of conv_mfma_kernel): bilinear `_upsample_add` residual, plain
    residual and a row-periodic one, against torch in fp64."""
    from lanemapping_amd import ops
    g = torch.Generator().manual_seed(cin + cout + h)
    x = torch.randn(B, cin, h, w, generator=g)
    wt = torch.randn(cout, cin, 1, 1, generator=g) / cin ** 0.5
    bias = torch.randn(cout, generator=g)
    ref = F.conv2d(x.double(), wt.double(), bias.double())
    xd = ops.to_nhwc(x.to(dev)) if hasattr(ops, 'to_nhwc') else x.to(dev).contiguous(memory_format=torch.channels_last)
    wp = ops.pack_mfma(wt.to(dev))
    if mode == 'up':
        coarse = torch.randn(B, cout, (h + 1) // 2, (w + 1) // 2, generator=g)
        ref = ref + F.interpolate(coarse.double(), size=(h, w), mode='bilinear', align_corners=True)
        y = ops.conv_mfma(xd, wp, cout, shift=bias.to(dev), res_up=coarse.to(dev).contiguous(memory_format=torch.channels_last))
    elif mode == 'res':
        r = torch.randn(B, cout, h, w, generator=g)
        ref = F.relu(ref + r.double())
        y = ops.conv_mfma(xd, wp, cout, shift=bias.to(dev), res=r.to(dev).contiguous(memory_format=torch.channels_last), act=ops.ACT_RELU)
    else:
        r = torch.randn(h * w, cout, generator=g)                 # one residual row per pixel of an image, shared by the batch
        ref = ref + r.double().t().reshape(1, cout, h, w)
        y = ops.conv_mfma(xd, wp, cout, shift=bias.to(dev), res=r.to(dev), res_rows=h * w)
    _close(y, ref.float(), 2e-5, f'1x1 {cin}->{cout} {mode}')


_LATERAL_AB = r"""
import sys, numpy as np, torch
from lanemapping_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(606)
out = {}
# latlayer2-shaped: 64 -> 256 + bilinear(coarse); latlayer1-shaped: 128 -> 256 + plain residual / + one residual row per pixel (res_rows).
# Sizes: whole 32-pixel tiles, ragged (B*H*W % 32 != 0, W % 32 != 0), fewer tiles than XCDs, more tiles than resident workgroups
for k, (B, h, w) in enumerate([(2, 96, 96), (1, 37, 53), (1, 5, 7), (3, 288, 288)]):
    x = torch.randn(B, 64, h, w, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(256, 64, 1, 1, generator=g) / 8).to(dev)
    bias = torch.randn(256, generator=g).to(dev)
    coarse = torch.randn(B, 256, (h + 1) // 2, (w + 1) // 2, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    out[f'up_{k}'] = ops.conv_mfma(x, ops.pack_mfma(wt), 256, shift=bias, res_up=coarse).permute(0, 2, 3, 1).cpu().numpy()
    x = torch.randn(B, 128, h, w, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    wt = (torch.randn(256, 128, 1, 1, generator=g) / 11).to(dev)
    r = torch.randn(B, 256, h, w, generator=g).to(dev).contiguous(memory_format=torch.channels_last)
    out[f'res_{k}'] = ops.conv_mfma(x, ops.pack_mfma(wt), 256, shift=bias, res=r).permute(0, 2, 3, 1).cpu().numpy()
    rr = torch.randn(h * w, 256, generator=g).to(dev)
    out[f'rows_{k}'] = ops.conv_mfma(x, ops.pack_mfma(wt), 256, shift=bias, res=rr, res_rows=h * w).permute(0, 2, 3, 1).cpu().numpy()
# strided views (what the FPN passes): input channels inside a wider tensor, residual and output as channel slices of wider tensors
torch.manual_seed(607)                  # (.normal_() below draws from the global generator: the same values in both processes)
wide = ops.new_act(2, 160, 64, 96, dev).normal_()
wt = (torch.randn(256, 64, 1, 1, generator=g) / 8).to(dev)
bias = torch.randn(256, generator=g).to(dev)
cw = ops.new_act(2, 320, 32, 48, dev).normal_()
ow = ops.new_act(2, 288, 64, 96, dev).zero_()
ops.conv_mfma(wide[:, 32:96], ops.pack_mfma(wt), 256, shift=bias, res_up=cw[:, 64:320], out=ow[:, 16:272])
out['up_views'] = ow.permute(0, 2, 3, 1).cpu().numpy()
wide = ops.new_act(2, 192, 32, 64, dev).normal_()
wt = (torch.randn(256, 128, 1, 1, generator=g) / 11).to(dev)
rw = ops.new_act(2, 288, 32, 64, dev).normal_()
ow = ops.new_act(2, 272, 32, 64, dev).zero_()
ops.conv_mfma(wide[:, 64:192], ops.pack_mfma(wt), 256, shift=bias, res=rw[:, 32:288], out=ow[:, 0:256])
out['res_views'] = ow.permute(0, 2, 3, 1).cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def test_lateral_kernel_bit_identical(dev, tmp_path):
    """lateral_mfma_kernel (round 6: the FPN's 1x1 laterals - weights resident in registers as the MFMA's A operand, pixels as B, no
    LDS / barrier, residual and stores straight from the accumulator quads, persistent workgroups) against the tiled conv_mfma_kernel
    (LM_CONV_LATERAL=0, read once per process) on the same inputs, bit for bit: bilinear coarse residual, plain residual, residual rows;
    ragged sizes (which fall back to the tiled kernel), fewer tiles than XCDs, more tiles than workgroups, strided channel-slice views."""
    import subprocess
    import sys
    outs = {}
    for flag in ('1', '0'):
        path = tmp_path / f'lateral_{flag}.npz'
        r = subprocess.run([sys.executable, '-c', _LATERAL_AB, str(path)], capture_output=True, text=True, timeout=900, cwd=ROOT,
                           env=dict(os.environ, LM_CONV_LATERAL=flag))
        assert r.returncode == 0, r.stderr[-2000:]
        outs[flag] = np.load(path)
    assert len(outs['1'].files) == 14
    for k in outs['1'].files:
        assert np.array_equal(outs['1'][k], outs['0'][k]), f'{k}: lateral_mfma_kernel differs from conv_mfma_kernel'


_STEM_AB = r"""
import sys, numpy as np, torch
from lanemapping_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(77)
w = (torch.randn(7, 7, 3, 64, generator=g) / 12).to(dev)
s, sh = (torch.rand(64, generator=g) + 0.5).to(dev), torch.randn(64, generator=g).to(dev)
out = {}
for k, shape in enumerate([(2, 96, 96), (1, 130, 75), (3, 33, 200), (40, 64, 64)]):          # (40 tiles of 4 x 4 blocks: > 512 workgroups of work)
    B, H, W = shape
    u8 = torch.randint(0, 256, (B, H, W, 3), generator=g, dtype=torch.uint8).to(dev)
    out[f'u8_{k}'] = ops.stem(u8, w, s, sh).permute(0, 2, 3, 1).cpu().numpy()
    f32 = (u8.permute(0, 3, 1, 2).float() / 255.0 + 0.01 * torch.randn(B, 3, H, W, generator=g).to(dev)).contiguous()
    out[f'f32_{k}'] = ops.stem(f32, w, s, sh).permute(0, 2, 3, 1).cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def test_stem_mfma_bit_identical_to_valu(dev, tmp_path):
    """stem_mfma_kernel (7x7 s2 stem on v_mfma_f32_32x32x2_f32: weights resident in VGPRs, zero-weight pads so that the two k of an MFMA
    are neighbours in LDS, persistent workgroups) against the VALU stem_kernel (LM_STEM_VALU=1, read once per process) on the same inputs,
    bit for bit: u8 HWC and f32 planar tiles, ragged sizes, more tiles than resident workgroups."""
    import subprocess
    import sys
    res = {}
    for tag, env in (('mfma', {}), ('valu', {'LM_STEM_VALU': '1'})):
        path = str(tmp_path / f'{tag}.npz')
        subprocess.run([sys.executable, '-c', _STEM_AB, path], check=True, env={**os.environ, **env, 'PYTHONPATH': ROOT}, cwd=ROOT)
        res[tag] = np.load(path)
    for k in res['mfma'].files:
        a, b = res['mfma'][k], res['valu'][k]
        assert np.isfinite(a).all() and a.shape == b.shape
        assert np.array_equal(a, b), (k, float(np.abs(a - b).max()))


_SMALL_AB = r"""
import sys, numpy as np, torch
from lanemapping_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(78)
out = {}
for k, (B, H, W, cout, stride, pre, act, ld) in enumerate([(2, 48, 48, 16, 1, False, ops.ACT_NONE, 16), (1, 50, 37, 16, 2, False, ops.ACT_RELU, 16),
                                                          (3, 33, 70, 8, 1, True, ops.ACT_NONE, 16), (2, 144, 144, 5, 2, False, ops.ACT_NONE, 24),
                                                          (24, 96, 96, 16, 1, False, ops.ACT_RELU, 16)]):
    wide = torch.randn(B, H, W, ld, generator=g).to(dev)
    x = wide[..., :16].permute(0, 3, 1, 2)                    # NHWC-stored, pixel stride ld
    w = ops.pack_small((torch.randn(cout, 16, 3, 3, generator=g) / 12).to(dev))
    sc = (torch.rand(cout, generator=g) + 0.5).to(dev) if k % 2 == 0 else None
    sh = torch.randn(cout, generator=g).to(dev)
    y = ops.conv_small(x, w, cout, 3, 3, stride, 1, scale=sc, shift=sh, pre_relu=pre, act=act)
    out[f'y{k}'] = y.permute(0, 2, 3, 1).cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def test_small_conv_mfma_bit_identical_to_valu(dev, tmp_path):
    """small_conv3x3_mfma_kernel (16-input-channel 3x3 convolutions of the head on v_mfma_f32_16x16x4_f32) against the VALU
    small_conv_kernel (LM_SMALL_CONV_VALU=1, read once per process), bit for bit: strides 1 / 2, ragged sizes, Cout < 16, scale / shift,
    pre-ReLU, ReLU, a channel slice of a wider tensor, more tiles than resident workgroups; and against torch."""
    import subprocess
    import sys
    res = {}
    for tag, env in (('mfma', {}), ('valu', {'LM_SMALL_CONV_VALU': '1'})):
        path = str(tmp_path / f'{tag}.npz')
        subprocess.run([sys.executable, '-c', _SMALL_AB, path], check=True, env={**os.environ, **env, 'PYTHONPATH': ROOT}, cwd=ROOT)
        res[tag] = np.load(path)
    for k in res['mfma'].files:
        a, b = res['mfma'][k], res['valu'][k]
        assert np.isfinite(a).all() and a.shape == b.shape
        assert np.array_equal(a, b), (k, float(np.abs(a - b).max()))
    from lanemapping_amd import ops
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 16, 40, 56, generator=g)
    w = torch.randn(16, 16, 3, 3, generator=g) / 12
    bias = torch.randn(16, generator=g)
    for stride in (1, 2):
        y = ops.conv_small(x.to(dev), ops.pack_small(w.to(dev)), 16, 3, 3, stride, 1, shift=bias.to(dev))
        _close(y, F.conv2d(x, w, bias, stride, 1), 1e-5, f'small conv 16->16 s{stride}')


_TOKENS_AB = r"""
import sys, numpy as np, torch
from lanemapping_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(79)
out = {}
for k, (B, Hr, P) in enumerate([(2, 144, 72), (1, 40, 20), (3, 25, 12)]):       # (Hr = 25: a ragged last block of token rows)
    seg = torch.randn(B, 1, 2 * Hr, 2 * Hr, generator=g).to(dev)
    row = torch.randn(B, Hr, Hr, 16, generator=g).to(dev).permute(0, 3, 1, 2)
    out[f'tok{k}'] = ops.head_tokens(seg, row, P, 2, 4, -0.37).cpu().numpy()
for k, (B, P, R, D, ld) in enumerate([(2, 72, 144, 100, 300), (1, 5, 37, 64, 200), (1, 3, 50, 104, 312)]):      # (M = 185: a ragged last block of rows)
    hid = torch.randn(B * P * R, ld, generator=g).to(dev)
    w2 = (torch.randn(23, D, generator=g) / 10).to(dev)
    b2 = torch.randn(23, generator=g).to(dev)
    for name, t in zip(('ext', 'cls', 'off'), ops.head_stage2(hid, D, w2, b2, B, P, R)):
        out[f'{name}{k}'] = t.cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def test_head_tokens_lds_bit_identical_to_gather(dev, tmp_path):
    """head_tokens_lds_kernel (the proposal window's source rows staged in LDS, one workgroup per (image, proposal, 24 token rows)) against
    the one-thread-per-token gather kernel (LM_HEAD_TOKENS_GATHER=1, read once per process): the same expression in the same order, bit for
    bit - the BASELINE shape, a small one and one with a ragged last block; proposals whose window leaves the map (conv-bias columns).
    Also head_stage2_lds_kernel (rows staged through LDS with coalesced loads) against the one-row-per-lane kernel (LM_HEAD_STAGE2_DIRECT=1)."""
    import subprocess
    import sys
    res = {}
    for tag, env in (('lds', {}), ('gather', {'LM_HEAD_TOKENS_GATHER': '1', 'LM_HEAD_STAGE2_DIRECT': '1'})):
        path = str(tmp_path / f'{tag}.npz')
        subprocess.run([sys.executable, '-c', _TOKENS_AB, path], check=True, env={**os.environ, **env, 'PYTHONPATH': ROOT}, cwd=ROOT)
        res[tag] = np.load(path)
    for k in res['lds'].files:
        a, b = res['lds'][k], res['gather'][k]
        assert np.isfinite(a).all() and a.shape == b.shape
        assert np.array_equal(a, b), (k, float(np.abs(a - b).max()))


_GNSUM_AB = r"""
import sys, numpy as np, torch
from lanemapping_amd import ops
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(91)
out = {}
# (B, C, Ho, Wo, source size of the up-sampled middle term, leading dimensions of the three terms, cout): the semantic branches' shape
# (128 channels, 288^2 <- 144^2, s2 / s3 channel slices of the merged 256-wide tensors, feature_layer 8 and output_layer_endp 1),
# ragged tiles, a scale below 1/2, 64 and 256 channels
for k, (B, C, Ho, Wo, Hm, Wm, lds, cout) in enumerate([(2, 128, 288, 288, 144, 144, (256, 256, 128), 8), (1, 128, 288, 288, 144, 144, (256, 256, 128), 1),
                                                        (1, 128, 75, 50, 37, 21, (128, 160, 128), 5), (2, 64, 56, 57, 20, 28, (64, 64, 96), 8),
                                                        (1, 256, 40, 72, 20, 36, (256, 256, 256), 3), (3, 32, 9, 17, 4, 8, (32, 48, 32), 2)]):
    terms = []
    for (h, w), ld in zip(((Ho, Wo), (Hm, Wm), (Ho, Wo)), lds):
        wide = (torch.randn(B, h, w, ld, generator=g) * 2 + 0.3).to(dev)
        x = wide[..., ld - C:].permute(0, 3, 1, 2)
        terms.append((x, ops.gn_stats(x.contiguous(memory_format=torch.channels_last))))
    gamma, beta = (torch.rand(C, generator=g) + 0.5).to(dev), torch.randn(C, generator=g).to(dev)
    w16 = ops.pack_small((torch.randn(cout, C, 1, 1, generator=g) / 8).to(dev))
    bias = torch.randn(cout, generator=g).to(dev)
    ysum, y1 = ops.gn_relu_upsample_sum(terms, gamma, beta, (Ho, Wo), proj=(w16, bias, cout))
    only = ops.gn_relu_upsample_sum(terms, gamma, beta, (Ho, Wo), proj=(w16, bias, cout), keep_sum=False)
    nob = ops.gn_relu_upsample_sum(terms, gamma, beta, (Ho, Wo), proj=(w16, None, cout), keep_sum=False)
    out[f'sum{k}'], out[f'y1_{k}'], out[f'only{k}'], out[f'nobias{k}'] = ysum.cpu().numpy(), y1.cpu().numpy(), only.cpu().numpy(), nob.cpu().numpy()
np.savez(sys.argv[1], **out)
"""


def test_gn_sum_three_terms_lds_bit_identical(dev, tmp_path):
    """gn_sum3_lds_kernel (round 5: s2 + s3 + s4 + the branch's 1x1 output layer with the up-sampled middle term staged in LDS, affines and
    projection weights once per thread) against gn_relu_upsample_sum_kernel<3, 5> (LM_GN_SUM_LDS=0, read once per process): the same
    arithmetic in the same order, bit for bit - the sum, the projection with and without bias, with and without writing the sum; the
    BASELINE shape with channel-slice operands, ragged tiles, 32 / 64 / 128 / 256 channels."""
    import subprocess
    import sys
    res = {}
    for tag, env in (('lds', {}), ('per_output', {'LM_GN_SUM_LDS': '0'})):
        path = str(tmp_path / f'{tag}.npz')
        subprocess.run([sys.executable, '-c', _GNSUM_AB, path], check=True, env={**os.environ, **env, 'PYTHONPATH': ROOT}, cwd=ROOT)
        res[tag] = np.load(path)
    for k in res['lds'].files:
        a, b = res['lds'][k], res['per_output'][k]
        assert np.isfinite(a).all() and a.shape == b.shape
        assert np.array_equal(a, b), (k, float(np.abs(a - b).max()))


# ----------------------------------------------------------------------------------------------- raster / ingest
def test_raster_vs_oracle_and_roundtrip(dev):
    from lanemapping_amd import ops
    from oracle import raster_ref
    n = 1 << 20
    pts = synth.las_points(33, n)
    kw = dict(quat=(0.9238795, 0.0, 0.0, 0.3826834), trans=(3.0, -2.0, 0.5), bev_img_offset=(-20.0, -30.0),
              img_reso=(0.05, 0.05), local_min_ele=-3.0, ele_reso=0.05)
    # place the synthetic tile-frame points into the "LAS" frame with the reference's forward formula
    ref_p = raster_ref.params(**kw)
    q = np.array(kw['quat'], dtype=np.float64)
    q /= np.linalg.norm(q)
    w, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
    local = pts[:, :3].astype(np.float64) + np.array([kw['bev_img_offset'][0], kw['bev_img_offset'][1], 0.0])
    world = (local @ R.T + np.array(kw['trans'])).astype(np.float32)
    rec = np.concatenate([world, pts[:, 3:4]], axis=1).astype(np.float32)
    want = raster_ref.raster(rec, ref_p)
    proj, u8 = ops.bev_raster(torch.from_numpy(rec).to(dev), ops.make_raster_params(**kw), want_u8=True)
    assert np.array_equal(u8.cpu().numpy(), want), 'rasteriser must match the C oracle bit for bit'
    assert torch.equal(proj.cpu(), torch.from_numpy(want.astype(np.float32) / np.float32(255.0)).permute(2, 0, 1))
    # determinism (scatter-max is order independent)
    proj2 = ops.bev_raster(torch.from_numpy(rec).to(dev), ops.make_raster_params(**kw))
    assert torch.equal(proj, proj2)
    # ragged batch: tile 0 = first 300k points, tile 1 = empty, tile 2 = the rest, different parameters per tile
    kw2 = dict(kw, quat=(1, 0, 0, 0), trans=(0, 0, 0), bev_img_offset=(-25.0, -35.0))
    offs = [0, 300000, 300000, n]
    pars = [ops.make_raster_params(**kw), ops.make_raster_params(**kw), ops.make_raster_params(**kw2)]
    _, bu8 = ops.bev_raster_batch(torch.from_numpy(rec).to(dev), offs, pars, want_u8=True)
    assert np.array_equal(bu8[0].cpu().numpy(), raster_ref.raster(rec[:300000], ref_p))
    assert int(bu8[1].max()) == 0
    assert np.array_equal(bu8[2].cpu().numpy(), raster_ref.raster(rec[300000:], raster_ref.params(**kw2)))
    # round trip through the reference's inverse transform: every occupied pixel maps back to within one
    # pixel pitch / one elevation step of a point that fell into it
    occ = np.argwhere(want.sum(2) > 0)
    sel = occ[:: max(1, len(occ) // 200)]
    rows = np.floor((pts[:, 0].astype(np.float64)) / 0.05 + 0.5).astype(int)
    cols = np.floor((pts[:, 1].astype(np.float64)) / 0.05 + 0.5).astype(int)
    for r, c in sel:
        back = raster_ref.pixel_to_point(ref_p, r, c, want[r, c, 1])
        m = (rows == r) & (cols == c)
        assert m.any()
        d = np.abs(world[m].astype(np.float64) - back)
        assert d[:, :2].min(axis=0).max() <= 0.05 + 1e-3 and d[:, 2].min() <= 0.05 + 1e-3
    # empty input -> empty tile
    empty = ops.bev_raster(torch.zeros((0, 4), device=dev), ops.make_raster_params(**kw))
    assert float(empty.abs().max()) == 0.0


def test_tile_ingest(dev):
    from lanemapping_amd import ops
    u8 = synth.bev_tile_u8(5, 96)
    rgba = np.concatenate([u8, np.full((96, 96, 1), 255, np.uint8)], axis=2)
    out = ops.tile_ingest(torch.from_numpy(rgba[None]).to(dev))
    assert torch.equal(out.cpu()[0], torch.from_numpy(synth.bev_tile(5, 96)))


def test_torch_custom_ops_on_device(dev, net):
    """torch.ops.lanemap_hip.*: schema / fake-kernel consistency (torch.library.opcheck) of kernel-level ops on real tensors, and the
    stage ops the modules go through give the same bits as calling the module implementation directly."""
    from lanemapping_amd import ops, torch_ops
    x = ops.new_act(2, 64, 24, 20, dev).normal_()
    w = torch.randn(96, 64, 3, 3, device=dev)
    wp = ops.pack_mfma(w)
    sh = torch.randn(96, device=dev)
    args = (x, wp, 96, 3, 3, 1, 1, 1, None, sh, None, 1)
    torch.library.opcheck(torch.ops.lanemap_hip.conv2d_mfma.default, args, test_utils=('test_schema', 'test_faketensor'))
    assert torch.equal(torch.ops.lanemap_hip.conv2d_mfma(*args), ops.conv_mfma(x, wp, 96, 3, 3, 1, 1, 1, shift=sh, act=1))
    u8 = torch.randint(0, 255, (2, 64, 48, 3), device=dev, dtype=torch.uint8)
    torch.library.opcheck(torch.ops.lanemap_hip.tile_ingest.default, (u8,), test_utils=('test_schema', 'test_faketensor'))
    lg = torch.randn(2, 3, 64, 48, device=dev)
    torch.library.opcheck(torch.ops.lanemap_hip.decode_semantic.default, (lg, 0.2), test_utils=('test_schema', 'test_faketensor'))
    # stage ops == module implementations
    tiles = torch.from_numpy(synth.bev_batch([81], 1152)).to(dev)
    fpn = net.pcencoder.fpn
    with torch.no_grad():
        a = fpn(tiles)                                         # through torch.ops.lanemap_hip.fpn_encoder
        b = fpn._forward_impl(tiles)
        for u, v in zip(a, b):
            assert torch.equal(u, v)
        ya = net.backbone(a[0])
        assert torch.equal(ya, net.backbone._forward_impl(a[0]))
        ha = net.heads(ya, a[1], a[3])
        hb = net.heads._forward_impl(ya, a[1], a[3])
        for k in hb:
            assert torch.equal(ha[k], hb[k]), k
    with pytest.raises((NotImplementedError, RuntimeError)):
        fpn(tiles.cpu())                                       # no CPU kernel behind the op: refused, never a fallback


@pytest.mark.parametrize('case', ['constant', 'saturated', 'plateau', 'two_levels'])
def test_endp_topk_tied_scores(dev, case):
    """Tie-safe top-K: flat or saturated endpoint maps put far more than 4096 pixels on the threshold score; the reference's
    argsort never fails there, and the build's rule is "ties -> lower flat index".  lm_endp_topk returns exactly the K best under
    (score descending, index ascending), status 0, and the pipeline clusters them instead of raising."""
    from lanemapping_amd import ops, hostpost
    H = W = 256
    clip, K = 20, 512
    g = torch.Generator().manual_seed(5)
    if case == 'constant':
        x = torch.full((2, 1, H, W), -1.25)
    elif case == 'saturated':
        x = torch.full((2, 1, H, W), 30.0)                              # sigmoid == 1.0f everywhere
        x[1, 0, 100:140, 60:90] = -3.0
    elif case == 'plateau':
        x = torch.randn((2, 1, H, W), generator=g) - 4.0
        x[:, 0, 50:150, 30:200] = 2.5                                   # 17,000 tied pixels above everything else
    else:
        x = torch.full((2, 1, H, W), 0.5)
        x[0, 0, 40:44, 40:140] = 3.0                                    # 400 distinct-level pixels + ties at the lower level
        x[1, 0, 30:60, 30:60] = 3.0                                     # 900 > K
    idx, score, status = ops.endp_topk(x.to(dev), K=K, clip=clip)
    assert int(status.max()) == 0
    Hc, Wc = H - 2 * clip, W - 2 * clip
    for b in range(2):
        s = torch.sigmoid(x[b, 0, clip:H - clip, clip:W - clip].to(dev)).cpu().numpy().reshape(-1)      # the device's own fp32 sigmoid
        want = np.lexsort((np.arange(s.size), -s.astype(np.float64)))[:K]
        got = idx[b].cpu().numpy()
        assert np.array_equal(np.sort(got), np.sort(want)), f'{case}: wrong candidate set for tile {b}'
        assert np.array_equal(got, want), f'{case}: order (score desc, index asc)'
        pts, k_used = hostpost.cluster_endpoints(got, crop_w=Wc, clip=clip, k0=240, k_max=500)
        assert len(pts) >= 1 and k_used >= 240


@pytest.mark.parametrize('max_points,max_voxels', [(10, 100000), (3, 700)])
def test_voxelize_vs_oracle(dev, max_points, max_voxels):
    from lanemapping_amd import ops
    from oracle import lidar_ref
    lo, vs, grid = lidar_ref.voxel_geometry([-15., -25., -2., 15., 25., 2.], grid_shape=[96, 96, 10])
    pts = [synth.lidar_points(31, 60000), np.zeros((0, 4), np.float32), synth.lidar_points(32, 20000),
           np.array([[100., 0., 0., 1.]], np.float32)]                      # ragged batch: empty sample, all-outside sample
    f_ref, c_ref = lidar_ref.voxelize_ref(pts, lo, vs, grid, max_points, max_voxels)
    f, c, ends = ops.voxelize_batch([torch.from_numpy(p).to(dev) for p in pts], lo, vs, grid, max_points, max_voxels)
    assert c.shape[0] == c_ref.shape[0] and ends[1] == ends[0] and ends[3] == ends[2]
    assert np.array_equal(c.cpu().numpy(), c_ref)                           # same voxels in the same (first-appearance) order
    assert float((f[:, :4].cpu() - torch.from_numpy(f_ref)).abs().max()) <= 1e-6
    assert float(f[:, 4:].abs().max()) == 0.0
    # raster-order mode: the same voxel set (same cap), rows of every sample sorted by (z, y, x)
    f2, c2, ends2 = ops.voxelize_batch([torch.from_numpy(p).to(dev) for p in pts], lo, vs, grid, max_points, max_voxels,
                                       raster_order=True)
    assert ends2 == ends
    c2n, f2n = c2.cpu().numpy().astype(np.int64), f2.cpu().numpy()
    key_ref = ((c_ref[:, 0].astype(np.int64) * 64 + c_ref[:, 1]) * 4096 + c_ref[:, 2]) * 4096 + c_ref[:, 3]
    key2 = ((c2n[:, 0] * 64 + c2n[:, 1]) * 4096 + c2n[:, 2]) * 4096 + c2n[:, 3]
    assert np.all(np.diff(key2) > 0)
    order = np.argsort(key_ref)
    assert np.array_equal(key_ref[order], key2)
    assert float(np.abs(f2n[:, :4] - f_ref[order]).max()) <= 1e-6


def test_sparse_backbone_vs_oracle(dev):
    import cases
    from lanemapping_amd import ops
    from oracle import lidar_ref
    cfg = cases.small_lidar_cfg()
    m, sd = _lidar_module(dev, cfg)
    le = cfg.pcencoder['lidar_encoder']
    lo, vs, grid = lidar_ref.voxel_geometry(le['voxelize']['point_cloud_range'], grid_shape=le['voxelize']['grid_shape'])
    pts = [synth.lidar_points(41, 40000), synth.lidar_points(42, 25000)]
    f_ref, c_ref = lidar_ref.voxelize_ref(pts, lo, vs, grid, 10, 100000)
    bb = le['backnone']
    ref = lidar_ref.sparse_encoder_ref(f_ref, c_ref, 2, sd, 'lidar_modal_extractor.backbone.', bb['sparse_shape'],
                                       bb['encoder_channels'], bb['encoder_paddings'], bb['block_type'])
    feats = torch.zeros((f_ref.shape[0], 16))
    feats[:, :4] = torch.from_numpy(f_ref)
    with torch.no_grad():
        got = m.sparse_backbone(feats.to(dev), torch.from_numpy(c_ref).to(dev), 2, flip_h=False)
    assert float(ref.abs().max()) > 1e-2
    _close(got, ref, 1e-4, 'sparse encoder dense output')
    assert np.array_equal((got.cpu() != 0).any(dim=1).numpy(), (ref != 0).any(dim=1).numpy())   # same active sites


# ----------------------------------------------------------------------------------------------- f4: LAS ingest
@pytest.mark.parametrize('fmt,version,n', [(0, (1, 2), 1000), (1, (1, 2), 70001), (2, (1, 2), 255), (3, (1, 2), 256), (6, (1, 4), 4097)])
def test_las_read_vs_oracle(dev, tmp_path, fmt, version, n):
    """disk -> HBM LAS decode vs the numpy reader (read_las arithmetic in float64, cast to float32): bit-exact."""
    from lanemapping_amd import las_io
    from oracle import las_ref
    rng = np.random.RandomState(fmt)
    xyz = rng.rand(n, 3) * [57.6, 57.6, 3] + [351200.0, 3433000.0, 10.0]
    inten = rng.randint(0, 65535, n)
    path = str(tmp_path / 't.las')
    las_ref.write_las(path, xyz, inten, point_format=fmt, version=version, offset=(351000.0, 3433000.0, 0.0), extra_bytes=fmt % 2)
    shift = [351200.0, 3433000.0, 10.0]
    for sh in (None, shift):
        ref = las_ref.read_las_ref(path, shift=sh).astype(np.float32)
        got = las_io.read_las(path, dev, shift=sh).cpu().numpy()
        assert got.shape == (n, 4) and np.array_equal(got, ref)
    raw, hdr = las_io.read_las_raw(path, dev, shift=shift)
    assert hdr['n_points'] == n and np.array_equal(raw.cpu().numpy(), las_ref.read_las_ref(path, shift=shift, normalise=False).astype(np.float32))


def test_raster_edge_cases(dev):
    """Rasteriser: every point outside the tile, a single point, all points in one pixel, intensity at the clip bounds."""
    from lanemapping_amd import ops
    from oracle import raster_ref
    par, rp = ops.make_raster_params(local_min_ele=-0.5, ele_reso=0.02), raster_ref.params(local_min_ele=-0.5, ele_reso=0.02)
    cases_ = {
        'outside': np.array([[-5., 3., 0., 900.], [100., 3., 0., 900.], [3., -0.1, 0., 900.]], np.float32),
        'single': np.array([[10.0, 20.0, 0.3, 20000.]], np.float32),
        'one_pixel': np.concatenate([np.full((5000, 2), 7.012, np.float32), np.linspace(-0.5, 4.0, 5000, dtype=np.float32)[:, None],
                                     np.linspace(0, 65535, 5000, dtype=np.float32)[:, None]], axis=1),
        'clip': np.array([[1., 1., 0., 0.], [2., 2., 0., 800.], [3., 3., 0., 33000.], [4., 4., 0., 65535.]], np.float32),
    }
    for name, pts in cases_.items():
        _, u8 = ops.bev_raster(torch.from_numpy(pts).to(dev), par, want_u8=True)
        assert np.array_equal(u8.cpu().numpy(), raster_ref.raster(pts, rp)), name
    _, u8 = ops.bev_raster(torch.from_numpy(cases_['outside']).to(dev), par, want_u8=True)
    assert int(u8.sum()) == 0


def test_c_abi_from_plain_c(dev, tmp_path):
    """A plain C program (tests/c_abi/smoke.c: no Python, no torch types) drives liblanemap_hip.so through
    include/lanemap_hip.h: rasteriser bit-exact vs the C oracle, MFMA convolution vs a scalar loop, error reporting."""
    import subprocess
    from lanemapping_amd._lib import LIB_PATH
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / 'c_abi_smoke')
    lib_dir = os.path.dirname(LIB_PATH)
    cmd = ['gcc', os.path.join(root, 'tests', 'c_abi', 'smoke.c'), os.path.join(root, 'oracle', 'raster_ref.c'), '-O2', '-ffp-contract=off',
           '-std=gnu11', '-I', os.path.join(root, 'include'), '-I', '/opt/rocm/include', '-D__HIP_PLATFORM_AMD__', '-L', lib_dir,
           '-llanemap_hip', '-L/opt/rocm/lib', '-lamdhip64', f'-Wl,-rpath,{lib_dir}', '-Wl,-rpath,/opt/rocm/lib', '-lm', '-o', exe]
    subprocess.run(cmd, check=True, capture_output=True, text=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    print(r.stdout)
    assert r.returncode == 0 and 'C-ABI smoke OK' in r.stdout, r.stdout + r.stderr



@pytest.mark.parametrize('B,cin,cout,H,W,Hr,Wr,k', [(2, 64, 256, 24, 28, 12, 14, 1), (1, 32, 64, 17, 9, 5, 4, 1), (1, 64, 128, 20, 20, 7, 20, 3)])
def test_conv_with_upsampled_residual(dev, B, cin, cout, H, W, Hr, Wr, k):
    """lm_conv2d_nhwc_mfma_resup_f32 (`_upsample_add` in the epilogue) == upsample kernel + conv with that residual, bit for bit, and
    == torch within fp32 tolerance."""
    from lanemapping_amd import ops
    g = torch.Generator().manual_seed(41)
    nhwc = lambda t: t.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    x, coarse = torch.randn(B, cin, H, W, generator=g), torch.randn(B, cout, Hr, Wr, generator=g)
    w = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    bias = torch.randn(cout, generator=g)
    wp = ops.pack_mfma(w.to(dev))
    xd, cd = nhwc(x), nhwc(coarse)
    fused = ops.conv_mfma(xd, wp, cout, k, k, 1, k // 2, shift=bias.to(dev), res_up=cd, act=ops.ACT_RELU)
    two = ops.conv_mfma(xd, wp, cout, k, k, 1, k // 2, shift=bias.to(dev), res=ops.upsample_nhwc(cd, (H, W)), act=ops.ACT_RELU)
    assert torch.equal(fused, two), float((fused - two).abs().max())
    want = F.relu(F.conv2d(x, w, bias, 1, k // 2) + F.interpolate(coarse, size=(H, W), mode='bilinear', align_corners=True))
    _close(fused, want, 2e-5, 'conv + upsampled residual')



@pytest.mark.parametrize('N', [12, 320, 321, 324, 352])
def test_attention_vs_torch(dev, N):
    """lm_attention_f32: the MFMA kernel (321..352 tokens, padded keys masked) and the VALU kernel (other lengths) vs torch."""
    from lanemapping_amd import ops
    B, heads, dh = 2, 16, 64
    g = torch.Generator().manual_seed(N)
    qkv = torch.randn((B * N, 3 * heads * dh), generator=g) * 1.5
    q, k, v = [z.reshape(B, N, heads, dh).transpose(1, 2).double() for z in qkv.chunk(3, dim=-1)]
    want = (torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5, dim=-1) @ v).transpose(1, 2).reshape(B * N, heads * dh).float()
    got = ops.attention(qkv.to(dev).contiguous(), B, N, heads, dh, dh ** -0.5)
    _close(got, want, 1e-5, f'attention N={N}')


@pytest.mark.parametrize('B,N,heads', [(3, 12, 16), (2, 7, 4)])
def test_attention_masked_vs_plain_on_compacted_tokens(dev, B, N, heads):
    """lm_attention_masked_f32 (RowRef: the keys of a batch element are its FLAGGED tokens, compacted in token order inside the kernel)
    against lm_attention_f32 run on the gathered valid rows alone: bitwise on the rows of valid tokens, for random masks, an all-set
    mask (== the plain kernel on everything) and an all-zero mask (no keys: the call must not fault; rows are unspecified)."""
    from lanemapping_amd import ops
    dh = 64
    g = torch.Generator().manual_seed(100 * B + N)
    qkv = torch.randn((B * N, 3 * heads * dh), generator=g).to(dev)
    scale = dh ** -0.5
    for trial in range(4):
        if trial == 0:
            valid = torch.ones((B, N), dtype=torch.int32)
        elif trial == 1:
            valid = torch.zeros((B, N), dtype=torch.int32)
        else:
            valid = (torch.rand((B, N), generator=g) > 0.4).to(torch.int32)
            valid[0, 0] = 1
        out = ops.attention(qkv, B, N, heads, dh, scale, valid=valid.to(dev).contiguous())
        torch.cuda.synchronize()
        for b in range(B):
            idx = torch.nonzero(valid[b]).flatten()
            if idx.numel() == 0:
                continue
            rows = qkv[b * N:(b + 1) * N][idx.to(dev)].contiguous()
            want = ops.attention(rows, 1, int(idx.numel()), heads, dh, scale)
            got = out[b * N:(b + 1) * N][idx.to(dev)]
            assert torch.equal(got, want), (trial, b, float((got - want).abs().max()))
        if trial == 0:
            assert torch.equal(out, ops.attention(qkv, B, N, heads, dh, scale))



@pytest.mark.parametrize('B,cin,cout,H,W,dil', [(2, 128, 128, 60, 64, 1), (1, 256, 200, 61, 75, 1), (2, 128, 64, 120, 130, 2),
                                                (1, 160, 256, 85, 187, 3), (1, 64, 64, 288, 288, 1), (1, 256, 256, 144, 144, 2),
                                                (2, 256, 512, 144, 144, 1), (1, 16, 70, 64, 300, 1)])
def test_conv_winograd44_bit_identical_to_twin(dev, B, cin, cout, H, W, dil):
    """lm_conv3x3_winograd44_f32 (Winograd F(4x4,3x3), exact fp32 MFMA, 36 xi split over the four waves by quadrant, no V / M tensor in
    HBM) produces the SAME BITS as its materialising twin (three plain kernels sharing its arithmetic helpers) on ragged sizes,
    dilations 1-3, one to sixteen channel units, channel counts that are not multiples of the 64-channel N tile, with BN scale / shift,
    residual and ReLU; both are within 1e-4 of the tensor scale of an fp64 convolution (the price of F(4x4)'s transform constants:
    profiles/r3_f44_numerics_study.txt); channel slices as operands; GroupNorm statistics from the epilogue; deterministic."""
    from lanemapping_amd import ops
    assert ops.wino44_supported(H, W, cin, dil)
    g = torch.Generator().manual_seed(B * 1000 + cin + H + dil)
    x = torch.randn((B, cin, H, W), generator=g)
    w = torch.randn((cout, cin, 3, 3), generator=g) / (cin * 9) ** 0.5
    scale, shift = torch.rand(cout, generator=g) + 0.5, torch.randn(cout, generator=g)
    res = torch.randn((B, cout, H, W), generator=g)
    xd = x.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    rd = res.to(dev).permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
    wu = ops.pack_wino44(w.to(dev))
    wf = ops.pack_wino44_fragments(wu)
    sd, bd = scale.to(dev), shift.to(dev)
    y0 = ops.conv_wino44_twin(xd, wu, cout, dil, scale=sd, shift=bd, res=rd, act=ops.ACT_RELU)
    y1 = ops.conv_wino44(xd, wf, cout, dil, scale=sd, shift=bd, res=rd, act=ops.ACT_RELU)
    want = F.relu(F.conv2d(x.double(), w.double(), None, 1, dil, dil) * scale.double().view(1, -1, 1, 1) + shift.double().view(1, -1, 1, 1)
                  + res.double()).float()
    _close(y0, want, 1e-4, 'winograd F(4x4) twin vs fp64')
    _close(y1, want, 1e-4, 'winograd F(4x4) vs fp64')
    assert torch.equal(y0, y1), float((y0 - y1).abs().max())
    assert torch.equal(y1, ops.conv_wino44(xd, wf, cout, dil, scale=sd, shift=bd, res=rd, act=ops.ACT_RELU))      # deterministic
    # a channel slice of a wider tensor as input, and a channel slice as output; no scale / residual / activation
    wide = ops.new_act(B, cin + 32, H, W, dev).normal_()
    wide[:, 16:16 + cin].copy_(xd)
    outw = ops.new_act(B, cout + 8, H, W, dev).zero_()
    ops.conv_wino44(wide[:, 16:16 + cin], wf, cout, dil, shift=bd, out=outw[:, 4:4 + cout])
    y2 = ops.conv_wino44_twin(xd, wu, cout, dil, shift=bd)
    assert torch.equal(outw[:, 4:4 + cout], y2) and float(outw[:, :4].abs().max()) == 0 and float(outw[:, 4 + cout:].abs().max()) == 0
    if cout in (64, 128, 256):                             # (channel counts the standalone statistics kernel takes)
        y3, st = ops.conv_wino44(xd, wf, cout, dil, shift=bd, gn_eps=1e-5)
        assert torch.equal(y3, y2)
        _close(st, ops.gn_stats(y3, 1e-5), 2e-5, 'gn stats from the F(4x4) epilogue')
        y4, st2 = ops.conv_wino44(xd, wf, cout, dil, shift=bd, gn_eps=1e-5)
        assert torch.equal(st, st2)


def test_conv_winograd44_random_shapes_vs_twin(dev):
    """Thirty-six seeded random shapes through wino44_kernel and its twin: bit-identical, twice (the fused loop synchronises its single V
    buffer with a barrier in the middle of every slot and keeps late planes in registers across slots - a race would show as a run-to-run
    or kernel-to-twin difference on some shape).  Ragged sizes, dilation 1-3, 1-20 channel units, any Cout, with / without scale, residual,
    ReLU; image widths from the narrowest supported tile row up (11 tiles per row: a 32-tile block then spans four runs of adjacent tiles,
    the limit of the kernel's run table - every third shape is drawn from Tx = 11 .. 14)."""
    from lanemapping_amd import ops
    rng = np.random.RandomState(4404)
    done = 0
    while done < 36:
        dil = int(rng.choice([1, 1, 1, 2, 2, 3]))
        W = int(rng.randint(41 * dil, 57 * dil + 140)) if done % 3 else int(rng.randint(41 * dil, 57 * dil))     # every third: Tx = 11 .. 14
        H = int(rng.randint(5, 90))
        cin = 16 * int(rng.randint(1, 21))
        cout = int(rng.choice([rng.randint(1, 40), 64, 128, rng.randint(65, 300)]))
        B = int(rng.randint(1, 3))
        if not ops.wino44_supported(H, W, cin, dil):
            continue
        done += 1
        g = torch.Generator().manual_seed(1000 + done)
        x = ops.new_act(B, cin, H, W, dev)
        x.copy_(torch.randn((B, cin, H, W), generator=g).to(dev))
        w = (torch.randn((cout, cin, 3, 3), generator=g) / (cin * 9) ** 0.5).to(dev)
        use_scale, use_res, relu = bool(rng.randint(2)), bool(rng.randint(2)), bool(rng.randint(2))
        sc = (torch.rand(cout, generator=g) + 0.5).to(dev) if use_scale else None
        sh = torch.randn(cout, generator=g).to(dev)
        res = None
        if use_res:
            res = ops.new_act(B, cout, H, W, dev)
            res.copy_(torch.randn((B, cout, H, W), generator=g).to(dev))
        act = ops.ACT_RELU if relu else ops.ACT_NONE
        wu = ops.pack_wino44(w)
        wf = ops.pack_wino44_fragments(wu)
        y0 = ops.conv_wino44_twin(x, wu, cout, dil, scale=sc, shift=sh, res=res, act=act)
        y1 = ops.conv_wino44(x, wf, cout, dil, scale=sc, shift=sh, res=res, act=act)
        y2 = ops.conv_wino44(x, wf, cout, dil, scale=sc, shift=sh, res=res, act=act)
        tag = f'shape {done}: B{B} {cin}->{cout} {H}x{W} d{dil} scale={use_scale} res={use_res} relu={relu}'
        assert torch.equal(y0, y1), (tag, float((y0 - y1).abs().max()))
        assert torch.equal(y1, y2), tag
        if cin % 32 == 0:                       # (and against the direct MFMA kernel where it takes the shape: the twin shares the fused kernel's arithmetic)
            yd = ops.conv_mfma(x, ops.pack_mfma(w), cout, 3, 3, 1, dil, dil, scale=sc, shift=sh, res=res, act=act)
            _close(y1, yd, 1e-4, tag + ' vs direct')



def test_conv_winograd44_split_second_line(dev):
    """The split-precision SECOND LINE of wino44_kernel (LANEMAP_WINO_SPLIT=1; fp16 x 2 terms per fp32 operand, three v_mfma_f32_32x32x8_f16
    products, fp32 accumulation): bit-identical to its own materialising twin (same split helper, same three products in the same order) on
    seeded random shapes, deterministic, and as close to an fp64 convolution as the exact fp32 kernel is (the dropped lo x lo term is 2^-22
    of a product): the error of the split kernel may exceed the exact kernel's by at most 50 % + 2e-6 on every shape.  Also: U scaled by
    the layer's power of two is undone exactly (a layer whose weights are 2^-7 times smaller gives 2^-7 times the same bits), and inputs
    spanning six decades keep the error relative to the output scale."""
    from lanemapping_amd import ops
    rng = np.random.RandomState(6606)
    done = 0
    worst = 0.0
    while done < 12:
        dil = int(rng.choice([1, 1, 2, 3]))
        W = int(rng.randint(41 * dil, 57 * dil + 100))
        H = int(rng.randint(5, 70))
        cin = 16 * int(rng.choice([1, 4, 8, 16, 5]))
        cout = int(rng.choice([64, 128, 256, rng.randint(1, 200)]))
        B = int(rng.randint(1, 3))
        if not ops.wino44_supported(H, W, cin, dil):
            continue
        done += 1
        g = torch.Generator().manual_seed(2000 + done)
        amp = 10.0 ** rng.uniform(-3, 2)                     # activations from 1e-3 to 1e2
        x = ops.new_act(B, cin, H, W, dev)
        x.copy_((torch.randn((B, cin, H, W), generator=g) * amp).to(dev))
        w = (torch.randn((cout, cin, 3, 3), generator=g) / (cin * 9) ** 0.5).to(dev)
        use_scale, use_res, relu = bool(rng.randint(2)), bool(rng.randint(2)), bool(rng.randint(2))
        sc = (torch.rand(cout, generator=g) + 0.5).to(dev) if use_scale else None
        sh = (torch.randn(cout, generator=g) * amp).to(dev)
        res = None
        if use_res:
            res = ops.new_act(B, cout, H, W, dev)
            res.copy_((torch.randn((B, cout, H, W), generator=g) * amp).to(dev))
        act = ops.ACT_RELU if relu else ops.ACT_NONE
        wu = ops.pack_wino44(w)
        ws = ops.pack_wino44_fragments_split(wu)
        assert isinstance(ws, ops.SplitFragments) and ws.words.shape == ops.pack_wino44_fragments(wu).shape
        y0 = ops.conv_wino44_twin(x, wu, cout, dil, scale=sc, shift=sh, res=res, act=act, split=True)
        y1 = ops.conv_wino44(x, ws, cout, dil, scale=sc, shift=sh, res=res, act=act)
        y2 = ops.conv_wino44(x, ws, cout, dil, scale=sc, shift=sh, res=res, act=act)
        ye = ops.conv_wino44(x, ops.pack_wino44_fragments(wu), cout, dil, scale=sc, shift=sh, res=res, act=act)
        tag = f'split shape {done}: B{B} {cin}->{cout} {H}x{W} d{dil} amp={amp:.1e} scale={use_scale} res={use_res} relu={relu}'
        assert torch.equal(y0, y1), (tag, float((y0 - y1).abs().max()))
        assert torch.equal(y1, y2), tag
        ref = F.conv2d(x.double(), w.double(), None, 1, dil, dil)
        if sc is not None:
            ref = ref * sc.double().view(1, -1, 1, 1)
        ref = ref + sh.double().view(1, -1, 1, 1)
        if res is not None:
            ref = ref + res.double()
        if relu:
            ref = F.relu(ref)
        scale = max(1.0, float(ref.abs().max()))
        e_split, e_exact = float((y1.double() - ref).abs().max()) / scale, float((ye.double() - ref).abs().max()) / scale
        worst = max(worst, e_split / max(e_exact, 1e-9))
        print(f'{tag}: error / output scale: split {e_split:.2e}, exact fp32 {e_exact:.2e}')
        assert e_split <= 1.5 * e_exact + 2e-6, tag
        if done == 1:                                        # the power of two of U is undone exactly
            y3 = ops.conv_wino44(x, ops.pack_wino44_fragments_split(ops.pack_wino44(w * 2.0 ** -7)), cout, dil)
            y4 = ops.conv_wino44(x, ws, cout, dil)
            assert torch.equal(y3 * 2.0 ** 7, y4), tag
    print(f'worst split / exact error ratio over {done} shapes: {worst:.2f}')


@pytest.mark.parametrize('seed', [11, 12, 13, 14])
def test_raster_fuzz_vs_c_oracle(dev, seed):
    """Random tile geometry (rotation, non-unit quaternion, offsets, resolutions, tile size), ragged point counts that are
    not multiples of the 8192-point chunk, points partly outside: bit-exact u8 tiles and f32 = u8 / 255 vs the C oracle."""
    from lanemapping_amd import ops
    from oracle import raster_ref
    rng = np.random.RandomState(seed)
    H, W = 16 * rng.randint(4, 40), 16 * rng.randint(4, 40)
    B = rng.randint(1, 4)
    pars, rps, clouds = [], [], []
    for b in range(B):
        q = rng.randn(4) * [1.0, 0.05, 0.05, 0.3]
        q[0] = abs(q[0]) + 0.5
        kw = dict(quat=q, trans=rng.randn(3) * 5, bev_img_offset=rng.randn(2), img_reso=(0.04 + 0.03 * rng.rand(), 0.04 + 0.03 * rng.rand()),
                  local_min_ele=-1.0 + rng.rand(), ele_reso=0.01 + 0.03 * rng.rand())
        pars.append(ops.make_raster_params(**kw))
        rps.append(raster_ref.params(**kw))
        n = int(rng.choice([0, 1, 777, 8192, 8193, 50000 + rng.randint(0, 9000)]))
        uv = rng.rand(n, 2) * [H * kw['img_reso'][0] * 1.2, W * kw['img_reso'][1] * 1.2] - 0.1 + kw['bev_img_offset']
        tile_xyz = np.concatenate([uv, rng.rand(n, 1) * 3 - 1.0], axis=1)
        from oracle import img2pc_ref
        world = np.stack([img2pc_ref.rotate(q, p) for p in tile_xyz]) + kw['trans'] if n else np.zeros((0, 3))
        clouds.append(np.concatenate([world, rng.randint(0, 65535, (n, 1))], axis=1).astype(np.float32))
    offs = np.concatenate([[0], np.cumsum([len(c) for c in clouds])])
    allp = torch.from_numpy(np.concatenate(clouds) if sum(len(c) for c in clouds) else np.zeros((0, 4), np.float32)).to(dev)
    if allp.shape[0] == 0:
        allp = torch.zeros((1, 4), device=dev)[:0]
    out, u8 = ops.bev_raster_batch(allp.contiguous(), offs.tolist(), pars, H, W, want_u8=True)
    for b in range(B):
        want = raster_ref.raster(clouds[b], rps[b], H, W)
        assert np.array_equal(u8[b].cpu().numpy(), want), f'tile {b} of seed {seed}'
        assert np.array_equal(out[b].cpu().numpy(), (want.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1))


def test_pack_segments_vs_torch(dev):
    """lm_pack_segments (ops.pack_readback: the decode outputs of a batch gathered into one block for ONE device-to-host copy): random
    segment counts, sizes that are not multiples of 16 bytes, sources that are only 4-byte aligned, dtypes of every width the pipeline packs;
    the block must hold every segment bit for bit at its 256-byte aligned offset and leave the gaps alone."""
    from lanemapping_amd import ops
    rng = np.random.RandomState(77)
    for trial in range(12):
        nseg = int(rng.randint(1, 9))
        ts = []
        for _ in range(nseg):
            dt = [torch.float32, torch.float64, torch.int32, torch.float32][int(rng.randint(4))]
            n = int(rng.choice([1, 3, 4, 5, 17, 72 * 2, 512, 72 * 144, 1000003]))
            base = torch.randint(-2 ** 31, 2 ** 31 - 1, (n * (2 if dt == torch.float64 else 1) + 8,), dtype=torch.int32, device=dev)
            off = int(rng.randint(0, 3)) * (2 if dt == torch.float64 else 1)          # 4-byte (8 for f64) aligned, not 16
            ts.append(base[off:off + n * (2 if dt == torch.float64 else 1)].view(dt))
        sentinel = torch.full((sum(t.numel() * t.element_size() for t in ts) + 256 * nseg + 64,), 0xA5, dtype=torch.uint8, device=dev)
        block, segs = ops.pack_readback(ts, block=sentinel)
        assert block.data_ptr() == sentinel.data_ptr()
        torch.cuda.synchronize()
        hb = block.cpu().numpy()
        covered = np.zeros(hb.size, bool)
        for t, (o, nb) in zip(ts, segs):
            assert o % 256 == 0 and nb == t.numel() * t.element_size()
            assert np.array_equal(hb[o:o + nb], t.contiguous().view(torch.uint8).cpu().numpy()), (trial, o, nb)
            covered[o:o + nb] = True
        assert np.all(hb[~covered] == 0xA5)                                          # nothing written outside the segments


@pytest.mark.parametrize('n', [0, 1, 63, 4096, 4097, 70001, 5_000_011, 33_554_433])
def test_exclusive_scan_u32(dev, n):
    """lm_exclusive_scan_u32 (the library's own three-phase scan, csrc/prim.hip) against numpy's cumsum, in place and out of place,
    with wrap-around."""
    from lanemapping_amd import ops
    rng = np.random.default_rng(n + 7)
    x = rng.integers(0, 5 if n < 10 ** 6 else 2 ** 31, size=n, dtype=np.int64).astype(np.uint32)
    want = np.zeros(n, dtype=np.uint32)
    if n > 1:
        want[1:] = np.cumsum(x[:-1].astype(np.uint64)).astype(np.uint32)
    xd = torch.from_numpy(x.view(np.int32)).to(dev)
    got = ops.exclusive_scan_u32(xd).cpu().numpy().view(np.uint32)
    np.testing.assert_array_equal(got, want)
    ops.exclusive_scan_u32(xd, out=xd)
    np.testing.assert_array_equal(xd.cpu().numpy().view(np.uint32), want)


@pytest.mark.parametrize('n,end_bit,kind', [(0, 32, 'rand'), (1, 32, 'rand'), (64, 8, 'rand'), (4097, 32, 'rand'), (100_003, 24, 'few'),
                                            (1_000_000, 32, 'rand'), (4_194_304, 24, 'cells'), (3_000_001, 16, 'equal'),
                                            (200_001, 13, 'rand'), (70_000, 3, 'rand'), (500_000, 23, 'cells')])
def test_sort_pairs_u32_stable(dev, n, end_bit, kind):
    """lm_sort_pairs_u32 (LSD radix sort, csrc/prim.hip) = numpy's STABLE argsort on the masked keys: duplicates keep their input
    order (the voxeliser numbers voxels by first point and keeps the first max_points points, so stability is the contract).  Exactly
    the low end_bit bits take part - also when end_bit is not a multiple of the 8-bit digit - so keys may carry payload above them, and
    the all-ones invalid key still sorts last."""
    from lanemapping_amd import ops
    rng = np.random.default_rng(n + end_bit)
    if kind == 'rand':
        k = rng.integers(0, 2 ** 32, size=n, dtype=np.uint64).astype(np.uint32)
    elif kind == 'few':
        k = rng.integers(0, 37, size=n).astype(np.uint32) * 449
    elif kind == 'equal':
        k = np.full(n, 12345, dtype=np.uint32)
    else:      # voxel cells with 10 % invalid points
        k = rng.integers(0, 600 * 600 * 21, size=n).astype(np.uint32)
        k[rng.random(n) < 0.1] = 0xFFFFFFFF
    v = np.arange(n, dtype=np.uint32)
    mask = np.uint32(0xFFFFFFFF) if end_bit >= 32 else np.uint32((1 << end_bit) - 1)
    order = np.argsort(k & mask, kind='stable')
    kd, vd = torch.from_numpy(k.view(np.int32)).to(dev), torch.from_numpy(v.view(np.int32)).to(dev)
    ops.sort_pairs_u32_(kd, vd, end_bit)
    np.testing.assert_array_equal(vd.cpu().numpy().view(np.uint32), v[order])
    np.testing.assert_array_equal(kd.cpu().numpy().view(np.uint32), k[order])



# ----------------------------------------------------------------------------------------------- round 3: parity at the bench's own sizes
def test_raster_headline_launch_vs_oracle(dev):
    """The rasteriser at the HEADLINE shape - one u8_only launch of 16 tiles x 4,194,304 points (512 partition chunks per tile, a full
    16-entry argument block) - against oracle/raster_ref.c on the first and the last tile of the launch: a wrong record slot at a chunk
    index >= 128, or in tile 15's argument entry, would pass every smaller test.  Tiles differ (4 distinct clouds, different
    elevation windows per tile), so a tile written into the wrong slot shows as well."""
    from lanemapping_amd import ops
    from oracle import raster_ref
    n, batch = 4194304, 16
    clouds = [synth.las_points(500 + i, n) for i in range(4)]
    points = torch.cat([torch.from_numpy(clouds[i % 4]) for i in range(batch)]).to(dev)
    offs = [i * n for i in range(batch + 1)]
    kws = [dict(local_min_ele=-0.5 - 0.01 * t, ele_reso=0.02) for t in range(batch)]
    out = torch.empty((batch, 1152, 1152, 3), device=dev, dtype=torch.uint8)
    ops.bev_raster_batch(points, offs, [ops.make_raster_params(**kw) for kw in kws], out_u8=out, u8_only=True)
    for t in (0, 15, 6):
        want = raster_ref.raster(clouds[t % 4], raster_ref.params(**kws[t]), 1152, 1152)
        assert np.array_equal(out[t].cpu().numpy(), want), f'tile {t} of the 16 x 4,194,304-point launch differs from the C oracle'
    # the same launch again into the same buffer: deterministic
    first = out.clone()
    ops.bev_raster_batch(points, offs, [ops.make_raster_params(**kw) for kw in kws], out_u8=out, u8_only=True)
    assert torch.equal(first, out)


def test_stage_ops_opcheck_and_functional_weights(dev, net):
    """The stage ops as the dispatcher sees them: torch.library.opcheck (schema incl. the declared mutation of fea_up_out / col, fake
    kernel vs real outputs) on all three, and the weights are REAL operands - handed other tensors than the module's own, the op
    computes with those (and the module's own weights are back in place afterwards)."""
    from lanemapping_amd import ops, torch_ops
    enc, vit, head = net.pcencoder.fpn, net.backbone, net.heads
    x = torch.from_numpy(synth.bev_batch([2021], 1152)).to(dev)
    we, ne = torch_ops.stage_weights(enc), torch_ops.stage_name(enc)
    wv, nv = torch_ops.stage_weights(vit), torch_ops.stage_name(vit)
    wh, nh = torch_ops.stage_weights(head), torch_ops.stage_name(head)
    utils = ('test_schema', 'test_faketensor')
    with torch.no_grad():
        up = ops.new_act(1, 8, 288, 288, dev)
        torch.library.opcheck(torch.ops.lanemap_hip.fpn_encoder.default, (x, up, we, ne), test_utils=utils)
        fea, bi, en = torch.ops.lanemap_hip.fpn_encoder(x, up, we, ne)
        torch.library.opcheck(torch.ops.lanemap_hip.vit_backbone.default, (fea, wv, nv), test_utils=utils)
        y = torch.ops.lanemap_hip.vit_backbone(fea, wv, nv)
        col = ops.new_act(1, 16, 288, 288, dev)
        col[:, 8:] = up
        torch.library.opcheck(torch.ops.lanemap_hip.colprop_head.default, (y, col, wh, nh), test_utils=utils)
        # functional use: other weights through the same op == a module that owns those weights
        w2 = [t.detach().clone() for t in wv]
        for t in w2:
            if t.dtype == torch.float32 and t.dim() >= 2:
                t.mul_(0.75)
        y2 = torch.ops.lanemap_hip.vit_backbone(fea, w2, nv)
        own = [t.detach().clone() for t in wv]
        try:
            for p_, t in zip(wv, w2):                   # the same weights loaded INTO the module: the reference result
                p_.copy_(t)                             # (in place on the parameter itself: bumps its version, the packed cache repacks)
            want2 = vit(fea)
        finally:                                        # (`net` is a session fixture shared by every test_gpu_* file)
            for p_, t in zip(wv, own):
                p_.copy_(t)
        assert torch.equal(y2, want2) and not torch.equal(y2, y)
        assert torch.equal(torch.ops.lanemap_hip.vit_backbone(fea, wv, nv), y), "the module's own weights are back in place"
        assert all(a is b for a, b in zip(torch_ops.stage_weights(vit), wv))
