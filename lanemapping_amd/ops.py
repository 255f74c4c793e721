"""Thin torch-tensor front end of the C-ABI (plumbing only: pointers, shapes, the current HIP stream).

Activation convention: tensors are *logically* NCHW (same shapes as the reference) but stored
channels-last (NHWC) — ``x.stride(1) == 1`` — so every kernel sees pixel-major rows with the channel
vector contiguous.  A channel slice of a wider NHWC buffer is a legal operand (its pixel stride ``ld``
is read from ``x.stride(3)``).  Every function raises if a tensor is not on a HIP device: there is no
CPU fallback.
"""
import ctypes as C
import math

import torch

from ._lib import lib, check, LmRasterParams, LanemapHipError

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2


def _stream():
    # raw current-stream handle straight from the C bindings: torch.cuda.current_stream() spends ~8 us per call in Python
    # device-index plumbing, and every launch asks for it
    return C.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise LanemapHipError('lanemapping_amd ops need tensors on an MI355X (HIP) device; no CPU fallback exists')
    if t.device.index != torch._C._cuda_getDevice():
        # the library launches on the current HIP device and _stream() is that device's current stream
        raise LanemapHipError(f'tensor on cuda:{t.device.index} but the current device is cuda:{torch._C._cuda_getDevice()}: '
                              'call torch.cuda.set_device() (one process drives one GPU)')
    return C.c_void_p(t.data_ptr())


def new_act(B, C_, H, W, device, dtype=torch.float32):
    """Fresh NHWC-stored activation, returned as a logical [B,C,H,W] view."""
    return torch.empty((B, H, W, C_), device=device, dtype=dtype).permute(0, 3, 1, 2)


def as_nhwc(x):
    """Return (tensor, ld): x itself if it is NHWC-stored (channel stride 1, pixel-major rows, pixel
    stride ld >= C), else an NHWC copy."""
    B, C_, H, W = x.shape
    sb, sc, sh, sw = x.stride()
    ok = (sc == 1 or C_ == 1) and sw >= C_ and (sh == W * sw or H == 1) and (sb == H * W * sw or B == 1)
    if not ok:
        x = x.permute(0, 2, 3, 1).contiguous().permute(0, 3, 1, 2)
        sw = C_
    return x, sw


# ------------------------------------------------------------------------------- weight packing
def fold_bn(bn, conv_bias=None, eps=1e-5):
    """Eval-mode BatchNorm as a per-channel (scale, shift) applied to the raw conv accumulator."""
    scale = bn.weight / torch.sqrt(bn.running_var + eps)
    shift = bn.bias - bn.running_mean * scale
    if conv_bias is not None:
        shift = shift + conv_bias * scale
    return scale.float().contiguous(), shift.float().contiguous()


def pack_mfma(w):
    """[Cout,Cin,KH,KW] (or [N,K] linear) -> [KH*KW, CoutP, Cin] with CoutP = Cout rounded up to 128."""
    if w.dim() == 2:
        w = w[:, :, None, None]
    if w.dim() == 3:
        w = w[:, :, :, None]
    co, ci, kh, kw = w.shape
    cop = (co + 127) // 128 * 128
    p = torch.zeros((kh * kw, cop, ci), device=w.device, dtype=torch.float32)
    p[:, :co, :] = w.permute(2, 3, 0, 1).reshape(kh * kw, co, ci)
    return p.contiguous()


_W44_G = [[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]]


def pack_wino44(w):
    """[Cout,Cin,3,3] -> Winograd F(4x4,3x3) weights U = G g G^T as [36, CoutP, Cin] (xi = 6 i + j), CoutP = Cout rounded up to 64.
    The product is formed in fp64 and rounded to fp32 once (G carries 1/6, 1/12, 1/24: csrc/conv_wino44.hip)."""
    co, ci, kh, kw = w.shape
    assert kh == 3 and kw == 3
    G = torch.tensor(_W44_G, device=w.device, dtype=torch.float64)
    U = torch.einsum('ij,ocjk,lk->iloc', G, w.double(), G).reshape(36, co, ci).float()
    cop = (co + 63) // 64 * 64
    p = torch.zeros((36, cop, ci), device=w.device, dtype=torch.float32)
    p[:, :co, :] = U
    return p.contiguous()


def pack_wino44_fragments(wu):
    """pack_wino44 output U [36, CoutP, Cin] -> the per-wave-fragment order of lm_conv3x3_winograd44_f32:
    [36][Cin/8][CoutP/32][lane 64][4] with lane = khalf * 32 + row, channel = u*8 + khalf*4 + e."""
    xi, cop, ci = wu.shape
    assert xi == 36 and cop % 64 == 0 and ci % 16 == 0
    t = wu.reshape(36, cop // 32, 32, ci // 8, 2, 4)                       # xi, nt, row, u, khalf, e
    return t.permute(0, 3, 1, 4, 2, 5).contiguous().reshape(36, ci // 8, cop // 32, 64, 4)


class SplitFragments:
    """U fragments of the split-precision second line (lm_conv3x3_winograd44_split_f32): `words` = the fp16 term words in fragment order
    (a float32 tensor holding bits, the shape pack_wino44_fragments gives), `post` = 1 / the power of two U was scaled by."""
    __slots__ = ('words', 'post')

    def __init__(self, words, post):
        self.words, self.post = words, float(post)

    @property
    def shape(self):
        return self.words.shape


def split_scale(wu):
    """The power of two that brings max |U| to [2^12, 2^13): the fp16 low terms of U then stay normal numbers."""
    m = float(wu.abs().max())
    return 1.0 if m == 0.0 else 2.0 ** (12 - math.floor(math.log2(m)))


def pack_wino44_fragments_split(wu):
    """pack_wino44 output U [36, CoutP, Cin] (on the GPU) -> SplitFragments: U * 2^k in fragment order, split into fp16 term words on the
    device by the helper the kernels use (lm_wino44_split_fragments)."""
    assert wu.is_cuda, 'the fragments are split on the device (no CPU path)'
    sc = split_scale(wu)
    frag = pack_wino44_fragments(wu * sc)
    out = torch.empty_like(frag)
    check(lib().lm_wino44_split_fragments(_stream(), _ptr(frag), _ptr(out), frag.numel() // 4))
    return SplitFragments(out, 1.0 / sc)


def pack_small(w):
    """[Cout<=16,Cin,KH,KW] -> [KH*KW, Cin, 16]."""
    co, ci, kh, kw = w.shape
    p = torch.zeros((kh * kw, ci, 16), device=w.device, dtype=torch.float32)
    p[:, :, :co] = w.permute(2, 3, 1, 0).reshape(kh * kw, ci, co)
    return p.contiguous()


# ------------------------------------------------------------------------------- convolutions / GEMM
_conv_hook = None   # optional profiler hook: called as hook(kind, algorithmic_flops, launch_fn[, executed_flops])


def set_conv_hook(fn):
    global _conv_hook
    _conv_hook = fn


def conv_mfma(x, wp, cout, kh=1, kw=1, stride=1, pad=0, dil=1, scale=None, shift=None, res=None, act=ACT_NONE,
              out=None, res_rows=0, res_up=None):
    """res_up: a COARSE [B,cout,Hr,Wr] map that is added through bilinear (align_corners) interpolation to the output size inside the
    epilogue (`_upsample_add` of the FPN) - the upsampled map is never written."""
    x, ldx = as_nhwc(x)
    B, cin, H, W = x.shape
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    Ho = (H + 2 * ph - dil * (kh - 1) - 1) // stride + 1
    Wo = (W + 2 * pw - dil * (kw - 1) - 1) // stride + 1
    if out is None:
        out = new_act(B, cout, Ho, Wo, x.device)
    out_, ldy = as_nhwc(out)
    assert out_.data_ptr() == out.data_ptr(), 'conv_mfma: `out` must already be NHWC-stored'
    ldr = 0
    if res is not None:
        if res.dim() == 4:
            res, ldr = as_nhwc(res)
        else:
            ldr = res.stride(0)
    if res_up is not None:
        assert res is None and res_rows == 0
        ru, ldu = as_nhwc(res_up)
        assert ru.shape[0] == B and ru.shape[1] == cout

    def launch():
        if res_up is not None:
            check(lib().lm_conv2d_nhwc_mfma_resup_f32(_stream(), _ptr(x), ldx, _ptr(wp), wp.shape[1], _ptr(scale), _ptr(shift),
                                                      _ptr(ru), ldu, ru.shape[2], ru.shape[3], _ptr(out), ldy, B, H, W, cin, cout,
                                                      kh, kw, stride, ph, pw, dil, act))
            return
        check(lib().lm_conv2d_nhwc_mfma_f32(_stream(), _ptr(x), ldx, _ptr(wp), wp.shape[1], _ptr(scale), _ptr(shift),
                                            _ptr(res), ldr, res_rows, _ptr(out), ldy, B, H, W, cin, cout, kh, kw,
                                            stride, ph, pw, dil, act))
    if _conv_hook is not None:
        _conv_hook(f'conv {cin}->{cout} k{kh}x{kw} s{stride} d{dil} @{H}x{W} B{B}', 2.0 * B * Ho * Wo * cout * cin * kh * kw, launch)
    else:
        launch()
    return out


def _hooked(kind, flops, launch, executed=None):
    if _conv_hook is not None:
        _conv_hook(kind, flops, launch, executed)
    else:
        launch()


def wino44_supported(H, W, cin, dil=1):
    return bool(lib().lm_winograd44_supported(int(H), int(W), int(cin), int(dil)))


def conv_wino44(x, wf, cout, dil=1, scale=None, shift=None, res=None, act=ACT_NONE, out=None, gn_eps=None, gn_split=1):
    """3x3 / stride 1 / pad = dil convolution via Winograd F(4x4,3x3), exact fp32 MFMA, no transformed tensors in HBM
    (csrc/conv_wino44.hip wino44_kernel: 0.5625x the matrix work of the F(2x2) kernels).  wf = pack_wino44_fragments(pack_wino44(w)).
    Bit-identical to conv_wino44_twin.  With gn_eps returns (y, stats)."""
    x, ldx = as_nhwc(x)
    B, cin, H, W = x.shape
    cop = wf.shape[2] * 32
    y = out if out is not None else new_act(B, cout, H, W, x.device)
    y_, ldy = as_nhwc(y)
    assert y_.data_ptr() == y.data_ptr(), 'conv_wino44: `out` must already be NHWC-stored'
    r, ldr = (None, 0) if res is None else as_nhwc(res)
    part = None
    if gn_eps is not None:
        part = torch.empty((B, lib().lm_winograd44_gn_chunks(H, W, dil), cout, 2), device=x.device, dtype=torch.float64)
    tiles = lib().lm_winograd44_tiles(B, H, W, dil)
    if isinstance(wf, SplitFragments):          # second line: fp16 x 2 split products (three MFMA terms per product: 3 x the algorithmic count executed)
        _hooked(f'wino44split {cin}->{cout} k3x3 d{dil} @{H}x{W} B{B}', 2.0 * B * H * W * cout * cin * 9,
                lambda: check(lib().lm_conv3x3_winograd44_split_f32(_stream(), _ptr(x), ldx, _ptr(wf.words), cop, _ptr(scale), _ptr(shift), _ptr(r),
                                                                    ldr, _ptr(y), ldy, B, H, W, cin, cout, dil, act, _ptr(part), wf.post)),
                3 * 2.0 * 36 * tiles * cin * cout)
    else:
        _hooked(f'wino44 {cin}->{cout} k3x3 d{dil} @{H}x{W} B{B}', 2.0 * B * H * W * cout * cin * 9,
                lambda: check(lib().lm_conv3x3_winograd44_f32(_stream(), _ptr(x), ldx, _ptr(wf), cop, _ptr(scale), _ptr(shift), _ptr(r), ldr,
                                                              _ptr(y), ldy, B, H, W, cin, cout, dil, act, _ptr(part))),
                2.0 * 36 * tiles * cin * cout)
    if gn_eps is None:
        return y
    if gn_split > 1:
        stats = torch.empty((gn_split, B, cout // gn_split, 2), device=x.device, dtype=torch.float32)
        check(lib().lm_gn_finalize_split(_stream(), _ptr(part), _ptr(stats), B, H * W, cout, part.shape[1], gn_eps, gn_split))
        return y, stats
    stats = torch.empty((B, cout, 2), device=x.device, dtype=torch.float32)
    check(lib().lm_gn_finalize(_stream(), _ptr(part), _ptr(stats), B, H * W, cout, part.shape[1], gn_eps))
    return y, stats


def conv_wino44_twin(x, wu, cout, dil=1, scale=None, shift=None, res=None, act=ACT_NONE, out=None, split=False):
    """The same convolution through the materialising twin (V and M tensors in HBM, three plain kernels): test infrastructure,
    bit-identical to conv_wino44.  wu = pack_wino44(w).  split: the twin of the split-precision second line (same scale rule as
    pack_wino44_fragments_split)."""
    x, ldx = as_nhwc(x)
    B, cin, H, W = x.shape
    cop = wu.shape[1]
    y = out if out is not None else new_act(B, cout, H, W, x.device)
    y_, ldy = as_nhwc(y)
    assert y_.data_ptr() == y.data_ptr(), 'conv_wino44_twin: `out` must already be NHWC-stored'
    r, ldr = (None, 0) if res is None else as_nhwc(res)
    need = lib().lm_winograd44_twin_workspace_bytes(B, H, W, cin, cop, dil)
    ws = torch.empty(need, device=x.device, dtype=torch.uint8)
    if split:
        sc = split_scale(wu)
        wus = (wu * sc).contiguous()
        check(lib().lm_conv3x3_winograd44_split_twin_f32(_stream(), _ptr(x), ldx, _ptr(wus), cop, _ptr(scale), _ptr(shift), _ptr(r), ldr,
                                                         _ptr(y), ldy, B, H, W, cin, cout, dil, act, _ptr(ws), need, 1.0 / sc))
        return y
    check(lib().lm_conv3x3_winograd44_twin_f32(_stream(), _ptr(x), ldx, _ptr(wu), cop, _ptr(scale), _ptr(shift), _ptr(r), ldr,
                                               _ptr(y), ldy, B, H, W, cin, cout, dil, act, _ptr(ws), need))
    return y


def conv_mfma_gnstats(x, wp, cout, kh, kw, stride, pad, dil, shift, eps=1e-5):
    """conv (+bias) and, from the same kernel, the GroupNorm(C,C) statistics of its output -> (y, stats [B,C,2])."""
    x, ldx = as_nhwc(x)
    B, cin, H, W = x.shape
    Ho = (H + 2 * pad - dil * (kh - 1) - 1) // stride + 1
    Wo = (W + 2 * pad - dil * (kw - 1) - 1) // stride + 1
    out = new_act(B, cout, Ho, Wo, x.device)
    nchunk = Ho * Wo // 64
    part = torch.empty((B, nchunk, cout, 2), device=x.device, dtype=torch.float64)
    stats = torch.empty((B, cout, 2), device=x.device, dtype=torch.float32)
    def launch():
        check(lib().lm_conv2d_nhwc_mfma_f32_gnstats(_stream(), _ptr(x), ldx, _ptr(wp), wp.shape[1], _ptr(shift), _ptr(out), cout,
                                                    _ptr(part), B, H, W, cin, cout, kh, kw, stride, pad, pad, dil))
    if _conv_hook is not None:
        _conv_hook(f'conv+gn {cin}->{cout} k{kh}x{kw} s{stride} d{dil} @{H}x{W} B{B}', 2.0 * B * Ho * Wo * cout * cin * kh * kw, launch)
    else:
        launch()
    check(lib().lm_gn_finalize(_stream(), _ptr(part), _ptr(stats), B, Ho * Wo, cout, nchunk, eps))
    return out, stats


def linear_mfma(x2d, wp, n_out, scale=None, shift=None, res=None, res_rows=0, act=ACT_NONE, out=None):
    """y[M,n_out] = act((x2d @ W^T) * scale + shift + res).  x2d [M,K] row-major (K % 32 == 0)."""
    assert x2d.dim() == 2 and x2d.stride(1) == 1
    M, K = x2d.shape
    if out is None:
        out = torch.empty((M, n_out), device=x2d.device, dtype=torch.float32)
    ldr = res.stride(0) if res is not None else 0
    def launch():
        check(lib().lm_conv2d_nhwc_mfma_f32(_stream(), _ptr(x2d), x2d.stride(0), _ptr(wp), wp.shape[1], _ptr(scale),
                                            _ptr(shift), _ptr(res), ldr, res_rows, _ptr(out), out.stride(0),
                                            1, 1, M, K, n_out, 1, 1, 1, 0, 0, 1, act))
    if _conv_hook is not None:
        _conv_hook(f'gemm M{M} K{K} N{n_out}', 2.0 * M * n_out * K, launch)
    else:
        launch()
    return out


def conv_small(x, w16, cout, kh=1, kw=1, stride=1, pad=0, scale=None, shift=None, pre_relu=False, act=ACT_NONE, out=None):
    x, ldx = as_nhwc(x)
    B, cin, H, W = x.shape
    ph, pw = (pad, pad) if isinstance(pad, int) else pad
    Ho = (H + 2 * ph - kh) // stride + 1
    Wo = (W + 2 * pw - kw) // stride + 1
    if out is None:
        out = new_act(B, cout, Ho, Wo, x.device)
    out_, ldy = as_nhwc(out)
    assert out_.data_ptr() == out.data_ptr()
    check(lib().lm_conv2d_nhwc_small(_stream(), _ptr(x), ldx, _ptr(w16), _ptr(scale), _ptr(shift), _ptr(out), ldy,
                                     B, H, W, cin, cout, kh, kw, stride, ph, pw, int(pre_relu), act))
    return out


def stem(x, w_k64, scale, shift):
    """proj -> relu(bn(conv7x7 s2)) [B,64,H/2,W/2] NHWC-stored.  proj: [B,3,H,W] f32 planar (the reference's tensor) or
    [B,H,W,3] uint8 (rasteriser / PNG reader output; u8 / 255 is applied inside the kernel: same bits)."""
    x = x.contiguous()
    if x.dtype == torch.uint8:
        B, H, W, C_ = x.shape
        assert C_ == 3, 'u8 tiles are [B,H,W,3]'
    else:
        B, _, H, W = x.shape
    Ho, Wo = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    out = new_act(B, 64, Ho, Wo, x.device)
    fn = lib().lm_stem_conv7x7_bn_relu_u8 if x.dtype == torch.uint8 else lib().lm_stem_conv7x7_bn_relu
    check(fn(_stream(), _ptr(x), _ptr(w_k64), _ptr(scale), _ptr(shift), _ptr(out), B, H, W))
    return out


def maxpool3x3s2(x):
    x, ld = as_nhwc(x)
    B, C_, H, W = x.shape
    assert ld == C_
    out = new_act(B, C_, (H - 1) // 2 + 1, (W - 1) // 2 + 1, x.device)
    check(lib().lm_maxpool3x3s2_nhwc(_stream(), _ptr(x), _ptr(out), B, H, W, C_))
    return out


# ------------------------------------------------------------------------------- norm / resize
def gn_stats(x, eps=1e-5):
    x, ld = as_nhwc(x)
    B, C_, H, W = x.shape
    assert ld == C_
    ws = torch.empty(lib().lm_gn_stats_workspace_bytes(B, H * W, C_) // 8, device=x.device, dtype=torch.float64)
    stats = torch.empty((B, C_, 2), device=x.device, dtype=torch.float32)
    check(lib().lm_gn_stats(_stream(), _ptr(x), _ptr(ws), _ptr(stats), B, H * W, C_, eps))
    return stats


def gn_relu_upsample(x, stats, gamma, beta, size, out=None, accumulate=False):
    x, ld = as_nhwc(x)
    B, C_, H, W = x.shape
    assert ld == C_
    Ho, Wo = size
    if out is None:
        assert not accumulate
        out = new_act(B, C_, Ho, Wo, x.device)
    check(lib().lm_gn_relu_upsample(_stream(), _ptr(x), _ptr(stats), _ptr(gamma), _ptr(beta), _ptr(out),
                                    B, H, W, Ho, Wo, C_, int(accumulate)))
    return out


def gn_relu_upsample_sum(terms, gamma, beta, size, out=None, proj=None, keep_sum=True):
    """((t0 + t1) + t2) with t_k = bilinear(relu(gn(x_k))) for terms = [(x_k, stats_k), ...] (at most 3) in one pass.
    proj = (w16, bias, cout[, out1]): also returns the 1x1 convolution of the sum (cout <= 8, weights packed by pack_small), computed
    from registers; with keep_sum=False the sum itself is never written and only the projection is returned."""
    n = len(terms)
    xs = [as_nhwc(x) for x, _ in terms]             # a term may be a channel slice of a wider NHWC tensor (ld > C)
    B, C_ = xs[0][0].shape[:2]
    for (x, ld) in xs:
        assert ld >= C_ and ld % 4 == 0 and x.shape[0] == B and x.shape[1] == C_
    Ho, Wo = size
    dev = xs[0][0].device
    if out is None and (keep_sum or proj is None):
        out = new_act(B, C_, Ho, Wo, dev)
    vp_arr, i_arr = C.c_void_p * n, C.c_int * n
    head = (_stream(), n, vp_arr(*[_ptr(x) for x, _ in xs]), vp_arr(*[_ptr(st) for _, st in terms]),
            i_arr(*[x.shape[2] for x, _ in xs]), i_arr(*[x.shape[3] for x, _ in xs]), i_arr(*[ld for _, ld in xs]),
            _ptr(gamma), _ptr(beta))
    if proj is None:
        check(lib().lm_gn_relu_upsample_sum(*head, _ptr(out), B, Ho, Wo, C_))
        return out
    w16, bias, cout = proj[:3]
    out1 = proj[3] if len(proj) > 3 and proj[3] is not None else new_act(B, cout, Ho, Wo, dev)
    o1, ldy1 = as_nhwc(out1)
    assert o1.data_ptr() == out1.data_ptr()
    check(lib().lm_gn_relu_upsample_sum_conv1x1(*head, _ptr(out) if keep_sum else None, B, Ho, Wo, C_, _ptr(w16), _ptr(bias), cout,
                                                _ptr(out1), ldy1))
    return (out, out1) if keep_sum else out1


def upsample_nhwc(x, size, add=None, out=None):
    x, ldx = as_nhwc(x)
    B, C_, H, W = x.shape
    Ho, Wo = size
    if out is None and add is None and (Ho, Wo) == (H, W):
        return x      # align_corners=True bilinear to the same size is the identity (source coordinate == destination, weight 1)
    if out is None:
        out = new_act(B, C_, Ho, Wo, x.device)
    out_, ldy = as_nhwc(out)
    assert out_.data_ptr() == out.data_ptr()
    lda = 0
    if add is not None:
        add, lda = as_nhwc(add)
    check(lib().lm_upsample_bilinear_nhwc(_stream(), _ptr(x), ldx, _ptr(add), lda, _ptr(out), ldy, B, H, W, Ho, Wo, C_))
    return out


def upsample_to_chw(x, size):
    x, ldx = as_nhwc(x)
    B, C_, H, W = x.shape
    Ho, Wo = size
    out = torch.empty((B, C_, Ho, Wo), device=x.device, dtype=torch.float32)
    check(lib().lm_upsample_bilinear_to_chw(_stream(), _ptr(x), ldx, _ptr(out), B, H, W, Ho, Wo, C_))
    return out


def layernorm(x2d, gamma, beta, eps=1e-5):
    assert x2d.is_contiguous()
    out = torch.empty_like(x2d)
    check(lib().lm_layernorm_rows(_stream(), _ptr(x2d), _ptr(gamma), _ptr(beta), _ptr(out), x2d.shape[0], x2d.shape[1], eps))
    return out


def unpatchify(tokens, B, G, P, C_):
    out = new_act(B, C_, G * P, G * P, tokens.device)
    check(lib().lm_unpatchify(_stream(), _ptr(tokens.contiguous()), _ptr(out), B, G, P, C_))
    return out


def attention(qkv, B, N, heads, dim_head, scale, valid=None):
    """valid (optional, [B, N] int32 on the device, N <= 64): per batch element only the flagged tokens are keys (compacted in token
    order: the arithmetic of a call on those tokens alone); every token still gets an output row."""
    assert qkv.is_contiguous()
    out = torch.empty((B * N, heads * dim_head), device=qkv.device, dtype=torch.float32)
    if valid is None:
        check(lib().lm_attention_f32(_stream(), _ptr(qkv), _ptr(out), B, N, heads, dim_head, float(scale)))
    else:
        assert valid.dtype == torch.int32 and valid.is_contiguous() and tuple(valid.shape) == (B, N)
        check(lib().lm_attention_masked_f32(_stream(), _ptr(qkv), _ptr(out), _ptr(valid), B, N, heads, dim_head, float(scale)))
    return out


# ------------------------------------------------------------------------------- head
def head_tokens(seg, row, P, prop_width, half_buff, seg_bias):
    """seg [B,1,288,288], row [B,16,144,144] (NHWC-stored) -> tok [B*P*144, 160]."""
    row, ld = as_nhwc(row)
    B, C_, Hr, Wr = row.shape
    assert C_ == 16 and ld == 16
    seg = seg.reshape(B, 2 * Hr, 2 * Wr).contiguous()
    tok = torch.empty((B * P * Hr, 160), device=row.device, dtype=torch.float32)
    check(lib().lm_head_tokens(_stream(), _ptr(seg), _ptr(row), _ptr(tok), float(seg_bias), B, P, Hr, Wr, prop_width, half_buff))
    return tok


def head_stage2(hid, D, w2, b2, B, P, R):
    M = hid.shape[0]
    ext2 = torch.empty((B, P, R, 3), device=hid.device, dtype=torch.float32)
    cls2 = torch.empty((B, P, R, 10), device=hid.device, dtype=torch.float32)
    off2 = torch.empty((B, P, R, 10), device=hid.device, dtype=torch.float32)
    check(lib().lm_head_stage2(_stream(), _ptr(hid), hid.stride(0), D, _ptr(w2), _ptr(b2), _ptr(ext2), _ptr(cls2), _ptr(off2), M))
    return ext2, cls2, off2


def head_proposal_conf(tok, wt, bias, B, P):
    L = tok.numel() // (B * P)
    conf = torch.empty((B, P, 2), device=tok.device, dtype=torch.float32)
    check(lib().lm_head_proposal_conf(_stream(), _ptr(tok), _ptr(wt), _ptr(bias), _ptr(conf), B * P, L))
    return conf


# ------------------------------------------------------------------------------- decode
def pack_readback(tensors, block=None):
    """Gather device tensors (contiguous, byte sizes multiples of 4) into ONE uint8 device block at 256-byte aligned offsets
    (lm_pack_segments); returns (block, [(offset, nbytes)]).  The pipeline then moves a batch's decode outputs to the host with one copy."""
    offs, sizes, off = [], [], 0
    for t in tensors:
        assert t.is_contiguous()
        nb = t.numel() * t.element_size()
        assert nb % 4 == 0, 'lm_pack_segments copies 4-byte words'
        off = (off + 255) // 256 * 256
        offs.append(off); sizes.append(nb)
        off += nb
    if block is None or block.numel() < off or block.device != tensors[0].device:
        block = torch.empty(max(off, 1), device=tensors[0].device, dtype=torch.uint8)
    n = len(tensors)
    src = (C.c_void_p * n)(*[t.data_ptr() for t in tensors])
    nby = (C.c_long * n)(*sizes)
    dof = (C.c_long * n)(*offs)
    check(lib().lm_pack_segments(_stream(), n, src, nby, dof, _ptr(block)))
    return block, list(zip(offs, sizes))


def decode_proposals(pconf, ext2, cls2, off2, exist_thre, prop_width, half_buff):
    B, P, R, _ = cls2.shape
    dev = cls2.device
    prop_conf = torch.empty((B, P, 2), device=dev, dtype=torch.float32)
    v_ext = torch.empty((B, P, R), device=dev, dtype=torch.float32)
    cls_conf = torch.empty((B, P, R, 10), device=dev, dtype=torch.float32)
    cls_idx = torch.empty((B, P, R), device=dev, dtype=torch.int32)
    cls_offset = torch.empty((B, P, R), device=dev, dtype=torch.float64)
    check(lib().lm_decode_proposals(_stream(), _ptr(pconf.contiguous()), _ptr(ext2.contiguous()), _ptr(cls2.contiguous()),
                                    _ptr(off2.contiguous()), _ptr(prop_conf), _ptr(v_ext), _ptr(cls_conf), _ptr(cls_idx),
                                    _ptr(cls_offset), B, P, R, float(exist_thre), prop_width, half_buff))
    return prop_conf, v_ext, cls_conf, cls_idx, cls_offset


def decode_orient(orient_logits):
    x, ld = as_nhwc(orient_logits)
    B, C_, H, W = x.shape
    out = torch.empty((B, H, W), device=x.device, dtype=torch.uint8)
    check(lib().lm_decode_orient(_stream(), _ptr(x), ld, C_, _ptr(out), B * H * W))
    return out


def decode_semantic(logits_chw, thre, raw_mode=False, want_biseg=True):
    x = logits_chw.contiguous()
    B, C_, H, W = x.shape
    assert C_ == 3
    sem = torch.empty((B, H, W), device=x.device, dtype=torch.uint8)
    biseg = rows = None
    if want_biseg:
        biseg = torch.empty((B, H, W), device=x.device, dtype=torch.float32)
        rows = torch.empty((B, H // 8, W), device=x.device, dtype=torch.float32)
    check(lib().lm_decode_semantic(_stream(), _ptr(x), _ptr(sem), _ptr(biseg), _ptr(rows), B, H, W, float(thre), int(raw_mode)))
    return sem, biseg, rows


def endp_topk(endp_logits, K=512, clip=20):
    x = endp_logits.contiguous()
    B, C_, H, W = x.shape
    assert C_ == 1
    ws = torch.empty(lib().lm_endp_topk_workspace_bytes(B), device=x.device, dtype=torch.uint8)
    idx = torch.empty((B, K), device=x.device, dtype=torch.int32)
    score = torch.empty((B, K), device=x.device, dtype=torch.float32)
    status = torch.empty((B,), device=x.device, dtype=torch.int32)
    check(lib().lm_endp_topk(_stream(), _ptr(x), _ptr(ws), _ptr(idx), _ptr(score), _ptr(status), B, H, W, clip, K))
    return idx, score, status


# ------------------------------------------------------------------------------- raster / ingest
def make_raster_params(quat=(1, 0, 0, 0), trans=(0, 0, 0), bev_img_offset=(0, 0), img_reso=(0.05, 0.05),
                       local_min_ele=0.0, ele_reso=0.05, inten_lo=800.0, inten_hi=33000.0):
    p = LmRasterParams()
    p.quat[:] = [float(v) for v in quat]
    p.trans[:] = [float(v) for v in trans]
    p.bev_img_offset[:] = [float(v) for v in bev_img_offset]
    p.img_reso[:] = [float(v) for v in img_reso]
    p.local_min_ele, p.ele_reso, p.inten_lo, p.inten_hi = float(local_min_ele), float(ele_reso), float(inten_lo), float(inten_hi)
    return p


_raster_ws = {}


def bev_raster_batch(points, tile_offsets, params, H=1152, W=1152, out=None, want_u8=False, u8_only=False, out_u8=None):
    """points [sum N,4] f32 (x,y,z,raw intensity) on device, tile_offsets: B+1 ints, params: list of LmRasterParams
    -> proj [B,3,H,W] f32 (= u8/255), optional u8 [B,H,W,3].  u8_only: only the u8 HWC tile is written (the stem takes it
    directly, ops.stem) - a quarter of the output bytes."""
    assert points.dim() == 2 and points.shape[1] == 4 and points.is_contiguous() and points.dtype == torch.float32
    B = len(params)
    assert len(tile_offsets) == B + 1
    offs = (C.c_long * (B + 1))(*[int(o) for o in tile_offsets])
    par = (LmRasterParams * B)(*params)
    cap = max([offs[b + 1] - offs[b] for b in range(B)] + [0])
    need = lib().lm_bev_raster_workspace_bytes(B, cap, H, W)
    key = (points.device, _stream().value)
    ws = _raster_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = _raster_ws[key] = torch.empty(need, device=points.device, dtype=torch.uint8)
    want_u8 = want_u8 or u8_only or out_u8 is not None
    if out is None and not u8_only:
        out = torch.empty((B, 3, H, W), device=points.device, dtype=torch.float32)
    u8 = None
    if want_u8:
        u8 = out_u8 if out_u8 is not None else torch.empty((B, H, W, 3), device=points.device, dtype=torch.uint8)
        assert u8.dtype == torch.uint8 and u8.is_contiguous() and tuple(u8.shape) == (B, H, W, 3)
    check(lib().lm_bev_raster_batch(_stream(), _ptr(points) if points.numel() else None, offs, par, B, _ptr(ws), ws.numel(),
                                    None if u8_only else _ptr(out), _ptr(u8), H, W))
    if u8_only:
        return u8
    return (out, u8) if want_u8 else out


def bev_raster(points, params, H=1152, W=1152, want_u8=False):
    """Single tile convenience wrapper: points [N,4] -> proj [3,H,W] (and u8 [H,W,3])."""
    r = bev_raster_batch(points, [0, points.shape[0]], [params], H, W, want_u8=want_u8)
    return (r[0][0], r[1][0]) if want_u8 else r[0]


def tile_ingest(u8_hwc):
    """[B,H,W,C>=3] uint8 (decoded PNG) -> [B,3,H,W] f32 = u8/255 (reference load_img contract)."""
    x = u8_hwc.contiguous()
    B, H, W, C_ = x.shape
    out = torch.empty((B, 3, H, W), device=x.device, dtype=torch.float32)
    check(lib().lm_tile_ingest_u8(_stream(), _ptr(x), _ptr(out), B, H, W, C_))
    return out


# ------------------------------------------------------------------------------- sparse-voxel LiDAR encoder (config 5)
_vox_ws = {}


def exclusive_scan_u32(x, out=None):
    """out[i] = x[0] + .. + x[i-1] over a device int32/uint32 vector (u32 arithmetic; csrc/prim.hip)."""
    x = x.contiguous()
    assert x.dim() == 1 and x.dtype in (torch.int32, torch.uint32)
    out = torch.empty_like(x) if out is None else out
    n = x.numel()
    ws = torch.empty((lib().lm_scan_workspace_bytes(n),), device=x.device, dtype=torch.uint8)
    check(lib().lm_exclusive_scan_u32(_stream(), _ptr(x), _ptr(out), n, _ptr(ws), ws.numel()))
    return out


def sort_pairs_u32_(keys, vals, end_bit=32):
    """Stable in-place sort of (keys, vals) (device int32 vectors read as u32) by the low end_bit bits of the keys (csrc/prim.hip)."""
    assert keys.dim() == 1 and keys.shape == vals.shape and keys.is_contiguous() and vals.is_contiguous()
    assert keys.dtype in (torch.int32, torch.uint32) and vals.dtype in (torch.int32, torch.uint32)
    n = keys.numel()
    ws = torch.empty((lib().lm_sort_pairs_workspace_bytes(n),), device=keys.device, dtype=torch.uint8)
    check(lib().lm_sort_pairs_u32(_stream(), _ptr(keys), _ptr(vals), n, int(end_bit), _ptr(ws), ws.numel()))
    return keys, vals


def voxelize_batch(points, range_lo, voxel_size, grid_xyz, max_points, max_voxels, ldf=16, raster_order=False):
    """Hard-voxelise a list of [N_i,4] device tensors -> (feats [V,ldf] (mean x,y,z,i; rest 0), coords [V,4] i32 (b,z,y,x),
    row_ends list).  One host sync at the end (the row count sizes every later launch).  Rows of a sample come in the
    reference's order (voxels numbered by first point) or, with raster_order, sorted by (z, y, x) - same voxel set."""
    dev = points[0].device
    B = len(points)
    cells = int(grid_xyz[0]) * int(grid_xyz[1]) * int(grid_xyz[2])
    caps = [min(int(p.shape[0]), int(max_voxels), cells) for p in points]
    cap = max(1, sum(caps))
    feats = torch.empty((cap, ldf), device=dev, dtype=torch.float32)
    coords = torch.empty((cap, 4), device=dev, dtype=torch.int32)
    ends = torch.zeros((B,), device=dev, dtype=torch.int32)
    nmax = max(int(p.shape[0]) for p in points)
    need = lib().lm_voxelize_workspace_bytes(nmax)
    key = (dev.index, _stream().value)
    ws = _vox_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty((need,), device=dev, dtype=torch.uint8)
        _vox_ws[key] = ws
    lo = (C.c_float * 3)(*[float(v) for v in range_lo])
    vs = (C.c_float * 3)(*[float(v) for v in voxel_size])
    g = (C.c_int * 3)(*[int(v) for v in grid_xyz])
    for b, p in enumerate(points):
        if not p.is_cuda:
            raise LanemapHipError('voxelize_batch needs device tensors; no CPU fallback exists')
        p = p.contiguous().float()
        assert p.dim() == 2 and p.shape[1] == 4
        base = C.c_void_p(ends[b - 1:b].data_ptr()) if b > 0 else None
        check(lib().lm_voxelize_hard(_stream(), _ptr(p) if p.shape[0] else None, p.shape[0], lo, vs, g, int(max_points),
                                     int(max_voxels), b, base, cap, _ptr(feats), ldf, _ptr(coords),
                                     C.c_void_p(ends[b:b + 1].data_ptr()), int(raster_order), _ptr(ws), ws.numel()))
    row_ends = [int(v) for v in ends.cpu()]
    V = row_ends[-1]
    return feats[:V], coords[:V], row_ends


def _ksp(kernel, stride, padding):
    return (C.c_int * 9)(*[int(v) for v in (*kernel, *stride, *padding)])


def sparse_grid(coords, B, shape):
    D, H, W = shape
    grid = torch.empty((B, D, H, W), device=coords.device, dtype=torch.int32)
    check(lib().lm_sparse_grid_build(_stream(), _ptr(coords), coords.shape[0], _ptr(grid), B, D, H, W))
    return grid


def sparse_conv_outputs(in_coords, B, in_shape, kernel, stride, padding):
    """Active output sites of a SparseConv3d -> (out_grid [B,Do,Ho,Wo] i32, out_coords [Vo,4] i32, out_shape).  Host sync."""
    out_shape = tuple((in_shape[a] + 2 * padding[a] - kernel[a]) // stride[a] + 1 for a in range(3))
    Do, Ho, Wo = out_shape
    dev = in_coords.device
    cells = B * Do * Ho * Wo
    n_in = in_coords.shape[0]
    taps_per_axis = [(kernel[a] + stride[a] - 1) // stride[a] for a in range(3)]
    cap = max(1, min(cells, n_in * taps_per_axis[0] * taps_per_axis[1] * taps_per_axis[2]))
    grid = torch.empty((B, Do, Ho, Wo), device=dev, dtype=torch.int32)
    coords = torch.empty((cap, 4), device=dev, dtype=torch.int32)
    count = torch.zeros((1,), device=dev, dtype=torch.int32)
    need = lib().lm_sparse_conv_outputs_workspace_bytes(cells)
    ws = torch.empty((need,), device=dev, dtype=torch.uint8)
    check(lib().lm_sparse_conv_outputs(_stream(), _ptr(in_coords), n_in, B, _ksp(kernel, stride, padding), Do, Ho, Wo,
                                       _ptr(grid), _ptr(coords), cap, _ptr(count), _ptr(ws), need))
    n = int(count.item())
    if n > cap:
        raise LanemapHipError(f'sparse_conv_outputs: {n} active sites exceed the bound {cap}')
    return grid, coords[:n], out_shape


def sparse_rulebook(out_coords, in_grid, kernel, stride, padding):
    B, D, H, W = in_grid.shape
    taps = kernel[0] * kernel[1] * kernel[2]
    nbr = torch.empty((out_coords.shape[0], taps), device=out_coords.device, dtype=torch.int32)
    check(lib().lm_sparse_rulebook(_stream(), _ptr(out_coords), out_coords.shape[0], _ptr(in_grid), B, D, H, W,
                                   _ksp(kernel, stride, padding), _ptr(nbr)))
    return nbr


def pack_sparse(w):
    """spconv weight [kD,kH,kW,Cin,Cout] -> kernel layout of lm_conv_gather_mfma_f32: Cin > 16: [taps, CoutP, CinP]
    (CinP = Cin up to 32s); Cin <= 16: tap pairs [ceil(taps/2), CoutP, 32] with k = (tap & 1) * 16 + c.  Zero filled."""
    kd, kh, kw, ci, co = w.shape
    taps = kd * kh * kw
    cop = (co + 127) // 128 * 128
    wt = w.reshape(taps, ci, co).permute(0, 2, 1).float()          # [taps, co, ci]
    if ci <= 16:
        p = torch.zeros(((taps + 1) // 2 * 2, cop, 16), device=w.device, dtype=torch.float32)
        p[:taps, :co, :ci] = wt
        return p.reshape(-1, 2, cop, 16).permute(0, 2, 1, 3).reshape(-1, cop, 32).contiguous()
    cip = (ci + 31) // 32 * 32
    p = torch.zeros((taps, cop, cip), device=w.device, dtype=torch.float32)
    p[:, :co, :ci] = wt
    return p.contiguous()


def sparse_ld(c):
    """Row stride of a c-channel sparse feature matrix: 16 for <= 16 channels (tap-pair kernel), else c up to 32s."""
    return 16 if c <= 16 else (c + 31) // 32 * 32


def conv_gather(x, nbr, wp, cin, cout, scale=None, shift=None, res=None, act=ACT_NONE):
    """Rulebook convolution: y[m, :cout] = act(bn(sum_t W[t] x[nbr[m, t]]) + res[m]).  x [V, sparse_ld(cin)] with zero padding
    channels; returns y [M, sparse_ld(cout)] padded the same way."""
    M, taps = nbr.shape
    cin_k = sparse_ld(cin)
    assert x.stride(1) == 1 and x.stride(0) >= cin_k
    ldy = sparse_ld(cout)
    y = torch.zeros((M, ldy), device=x.device, dtype=torch.float32) if ldy != cout else \
        torch.empty((M, ldy), device=x.device, dtype=torch.float32)
    def launch():
        check(lib().lm_conv_gather_mfma_f32(_stream(), _ptr(x), x.stride(0), _ptr(nbr), taps, _ptr(wp), wp.shape[1], _ptr(scale),
                                            _ptr(shift), _ptr(res), res.stride(0) if res is not None else 0, _ptr(y), ldy,
                                            M, cin_k, cout, act))
    if _conv_hook is not None:
        _conv_hook(f'spconv M{M} taps{taps} {cin}->{cout}', 2.0 * M * taps * cin * cout, launch)
    else:
        launch()
    return y


def sparse_to_dense(feats, coords, B, shape, C_, flip_h):
    """-> logical [B, C*D, H, W] tensor stored NHWC."""
    D, H, W = shape
    out = new_act(B, C_ * D, H, W, feats.device)
    check(lib().lm_sparse_to_dense_nhwc(_stream(), _ptr(feats), feats.stride(0), _ptr(coords), feats.shape[0], _ptr(out),
                                        B, D, H, W, C_, int(flip_h)))
    return out


def upsample_bicubic(x, size):
    x, ld = as_nhwc(x)
    B, C_, H, W = x.shape
    assert ld == C_
    out = new_act(B, C_, size[0], size[1], x.device)
    check(lib().lm_upsample_bicubic_nhwc(_stream(), _ptr(x), _ptr(out), B, H, W, C_, size[0], size[1]))
    return out
