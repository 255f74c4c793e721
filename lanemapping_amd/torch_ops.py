"""``torch.ops.lanemap_hip.*`` - the hot path as PyTorch custom ops (torch.library), over the SAME C-ABI library.

north_star asks for the path to be "exposed as PyTorch-ROCm custom ops" behind the reference's registries
(baseline/utils/registry.py:54-82 builds the modules; the modules call these ops).  Every op here is registered with a
schema and a fake (meta) kernel, so the dispatcher, ``torch.compile`` and ``torch.export`` see them as opaque device ops with
known output shapes.  There is NO second implementation: the device kernels are the ctypes calls of ``ops.py`` into
liblanemap_hip.so, registered for the ``cuda`` (HIP) device only - calling an op with CPU tensors fails in the dispatcher
("no kernel for CPU"), it never falls back.  The two host-side ops (endpoint clustering, polyline assembly) are host C++ of the
same library and are registered for ``cpu`` because that is where their inputs live.

Two levels:
  * stage ops - what the modules' ``forward`` go through: ``bev_raster``, ``fpn_encoder``, ``vit_backbone``, ``colprop_head``
    (+ ``endp_cluster``, ``polyline_assemble`` on the host).  A stage op takes the stage's WEIGHTS as a ``Tensor[]`` operand (every
    parameter and buffer of the module, in ``state_dict`` order - the tracer / ``torch.export`` see them as inputs of the op, and the
    op computes from the tensors it is handed) and the stage's STRUCTURE (which layers, strides, dilations) as a string: the name
    under which the module registered itself (``stage_name``: class name + a process-wide counter, never recycled).  Round 2 passed
    ``id(module)``: a process-local integer that could alias a recycled id and hid the weights from the graph.
  * kernel ops - one per device kernel family (``conv2d_mfma``, ``conv3x3_winograd44``, ``stem_conv7x7``, ``gn_relu_upsample``,
    ``layernorm_rows``, ``attention``, ``linear_mfma``, ``decode_proposals``, ``decode_semantic``, ``endp_topk``, ``tile_ingest``, ...).
Activations are logically NCHW, stored channels-last (ops.new_act); fake kernels return the same strides.
"""
import itertools
import weakref
from typing import List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from . import ops as _ops

_STAGES = weakref.WeakValueDictionary()      # stage name -> live module (the structure; the weights travel as op operands)
_STAGE_COUNTER = itertools.count()


def stage_name(module):
    """Name of a stage module for the stage ops: '<ClassName>#<n>', n from a process-wide counter (unique for the life of the process,
    unlike id(); two processes that build the same net in the same order agree on it)."""
    n = module.__dict__.get('_lanemap_stage_name')
    if n is None:
        n = module.__dict__['_lanemap_stage_name'] = f'{type(module).__name__}#{next(_STAGE_COUNTER)}'
    _STAGES[n] = module
    return n


def stage_weights(module):
    """Every parameter and buffer of the module, in state_dict order (cached slot list of PackedModule: no module walk per call)."""
    return [d[n] for d, n in module._slots() if d[n] is not None]


class _with_weights:
    """Run a stage with the weights the op was handed.  Fast path: they ARE the module's own tensors.  Otherwise (a functional caller,
    an exported graph replayed with other weights) they are swapped in for the duration of the call; the packed-weight cache is keyed
    by tensor identity + version, so it repacks by itself."""

    def __init__(self, spec, weights):
        m = _STAGES.get(spec)
        if m is None:
            raise RuntimeError(f'lanemap_hip: no live stage module registered as {spec!r}')
        slots = [(d, n) for d, n in m._slots() if d[n] is not None]
        if len(slots) != len(weights):
            raise RuntimeError(f'lanemap_hip: stage {spec} has {len(slots)} weight tensors, the op was handed {len(weights)}')
        self.m, self.swap = m, [(d, n, d[n], w) for (d, n), w in zip(slots, weights) if d[n] is not w]
        for d, n, own, w in self.swap:
            if own.shape != w.shape or own.dtype != w.dtype:
                raise RuntimeError(f'lanemap_hip: stage {spec}: weight {n} is {tuple(w.shape)} {w.dtype}, expected {tuple(own.shape)} {own.dtype}')

    def __enter__(self):
        for d, n, own, w in self.swap:
            d[n] = w if not isinstance(own, torch.nn.Parameter) or isinstance(w, torch.nn.Parameter) else torch.nn.Parameter(w, requires_grad=False)
        return self.m

    def __exit__(self, *exc):
        for d, n, own, w in self.swap:
            d[n] = own
        return False


def _stage(spec):
    m = _STAGES.get(spec)
    if m is None:
        raise RuntimeError(f'lanemap_hip: no live stage module registered as {spec!r}')
    return m


def _fake_act(like, B, C, H, W, dtype=torch.float32):
    return like.new_empty((B, H, W, C), dtype=dtype).permute(0, 3, 1, 2)


def _out_hw(H, W, kh, kw, stride, pad, dil):
    return (H + 2 * pad - dil * (kh - 1) - 1) // stride + 1, (W + 2 * pad - dil * (kw - 1) - 1) // stride + 1


def _define(name, fn, fake, mutates=(), device='cuda'):
    op = torch.library.custom_op(f'lanemap_hip::{name}', fn, mutates_args=mutates, device_types=device)
    op.register_fake(fake)
    return op


# ------------------------------------------------------------------------------------------------------------- kernel ops
def _conv2d_mfma(x: Tensor, w_packed: Tensor, cout: int, kh: int, kw: int, stride: int, pad: int, dil: int,
                 scale: Optional[Tensor], shift: Optional[Tensor], res: Optional[Tensor], act: int) -> Tensor:
    return _ops.conv_mfma(x, w_packed, cout, kh, kw, stride, pad, dil, scale=scale, shift=shift, res=res, act=act)


def _conv2d_mfma_fake(x, w_packed, cout, kh, kw, stride, pad, dil, scale, shift, res, act):
    Ho, Wo = _out_hw(x.shape[2], x.shape[3], kh, kw, stride, pad, dil)
    return _fake_act(x, x.shape[0], cout, Ho, Wo)


def _conv3x3_winograd44(x: Tensor, wu_frag: Tensor, cout: int, dil: int, scale: Optional[Tensor], shift: Optional[Tensor],
                        res: Optional[Tensor], act: int) -> Tensor:
    return _ops.conv_wino44(x, wu_frag, cout, dil, scale=scale, shift=shift, res=res, act=act)


def _conv3x3_fake(x, w, cout, dil, scale, shift, res, act):
    return _fake_act(x, x.shape[0], cout, x.shape[2], x.shape[3])


def _stem(x: Tensor, w_k64: Tensor, scale: Tensor, shift: Tensor) -> Tensor:
    return _ops.stem(x, w_k64, scale, shift)


def _stem_fake(x, w_k64, scale, shift):
    B, H, W = (x.shape[0], x.shape[1], x.shape[2]) if x.dtype == torch.uint8 else (x.shape[0], x.shape[2], x.shape[3])
    return _fake_act(x, B, 64, (H - 1) // 2 + 1, (W - 1) // 2 + 1)


def _maxpool(x: Tensor) -> Tensor:
    return _ops.maxpool3x3s2(x)


def _maxpool_fake(x):
    return _fake_act(x, x.shape[0], x.shape[1], (x.shape[2] - 1) // 2 + 1, (x.shape[3] - 1) // 2 + 1)


def _gn_stats(x: Tensor, eps: float) -> Tensor:
    return _ops.gn_stats(x, eps)


def _gn_stats_fake(x, eps):
    return x.new_empty((x.shape[0], x.shape[1], 2))


def _gn_relu_upsample(x: Tensor, stats: Tensor, gamma: Tensor, beta: Tensor, Ho: int, Wo: int) -> Tensor:
    return _ops.gn_relu_upsample(x, stats, gamma, beta, (Ho, Wo))


def _upsample_bilinear(x: Tensor, Ho: int, Wo: int) -> Tensor:
    y = _ops.upsample_nhwc(x, (Ho, Wo))
    return y.clone() if y.data_ptr() == x.data_ptr() else y          # (an op may not return its input)


def _resize_fake(x, *a):
    return _fake_act(x, x.shape[0], x.shape[1], a[-2], a[-1])


def _layernorm_rows(x: Tensor, gamma: Tensor, beta: Tensor, eps: float) -> Tensor:
    return _ops.layernorm(x, gamma, beta, eps)


def _attention(qkv: Tensor, B: int, N: int, heads: int, dim_head: int, scale: float) -> Tensor:
    return _ops.attention(qkv, B, N, heads, dim_head, scale)


def _linear_mfma(x: Tensor, w_packed: Tensor, n_out: int, scale: Optional[Tensor], shift: Optional[Tensor], res: Optional[Tensor],
                 res_rows: int, act: int) -> Tensor:
    return _ops.linear_mfma(x, w_packed, n_out, scale=scale, shift=shift, res=res, res_rows=res_rows, act=act)


def _tile_ingest(u8_hwc: Tensor) -> Tensor:
    return _ops.tile_ingest(u8_hwc)


def _decode_proposals(pconf: Tensor, ext2: Tensor, cls2: Tensor, off2: Tensor, exist_thre: float, prop_width: int,
                      half_buff: int) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    return _ops.decode_proposals(pconf, ext2, cls2, off2, exist_thre, prop_width, half_buff)


def _decode_proposals_fake(pconf, ext2, cls2, off2, exist_thre, prop_width, half_buff):
    B, P, R, _ = cls2.shape
    return (cls2.new_empty((B, P, 2)), cls2.new_empty((B, P, R)), cls2.new_empty((B, P, R, 10)),
            cls2.new_empty((B, P, R), dtype=torch.int32), cls2.new_empty((B, P, R), dtype=torch.float64))


def _decode_semantic(logits: Tensor, thre: float) -> Tuple[Tensor, Tensor, Tensor]:
    return _ops.decode_semantic(logits, thre)


def _decode_semantic_fake(logits, thre):
    B, _, H, W = logits.shape
    return logits.new_empty((B, H, W), dtype=torch.uint8), logits.new_empty((B, H, W)), logits.new_empty((B, H // 8, W))


def _decode_orient(orient_logits: Tensor) -> Tensor:
    return _ops.decode_orient(orient_logits)


def _endp_topk(endp_logits: Tensor, K: int, clip: int) -> Tuple[Tensor, Tensor, Tensor]:
    return _ops.endp_topk(endp_logits, K=K, clip=clip)


def _endp_topk_fake(endp_logits, K, clip):
    B = endp_logits.shape[0]
    return (endp_logits.new_empty((B, K), dtype=torch.int32), endp_logits.new_empty((B, K)), endp_logits.new_empty((B,), dtype=torch.int32))


conv2d_mfma = _define('conv2d_mfma', _conv2d_mfma, _conv2d_mfma_fake)
conv3x3_winograd44 = _define('conv3x3_winograd44', _conv3x3_winograd44, _conv3x3_fake)      # F(4x4,3x3): the FPN's 3x3 / stride-1 route
stem_conv7x7 = _define('stem_conv7x7', _stem, _stem_fake)
maxpool3x3s2 = _define('maxpool3x3s2', _maxpool, _maxpool_fake)
gn_stats = _define('gn_stats', _gn_stats, _gn_stats_fake)
gn_relu_upsample = _define('gn_relu_upsample', _gn_relu_upsample, lambda x, stats, gamma, beta, Ho, Wo: _resize_fake(x, Ho, Wo))
upsample_bilinear = _define('upsample_bilinear', _upsample_bilinear, lambda x, Ho, Wo: _resize_fake(x, Ho, Wo))
layernorm_rows = _define('layernorm_rows', _layernorm_rows, lambda x, gamma, beta, eps: torch.empty_like(x))
attention = _define('attention', _attention, lambda qkv, B, N, heads, dim_head, scale: qkv.new_empty((B * N, heads * dim_head)))
linear_mfma = _define('linear_mfma', _linear_mfma,
                      lambda x, w_packed, n_out, scale, shift, res, res_rows, act: x.new_empty((x.shape[0], n_out)))
tile_ingest = _define('tile_ingest', _tile_ingest,
                      lambda u8: u8.new_empty((u8.shape[0], 3, u8.shape[1], u8.shape[2]), dtype=torch.float32))
decode_proposals = _define('decode_proposals', _decode_proposals, _decode_proposals_fake)
decode_semantic = _define('decode_semantic', _decode_semantic, _decode_semantic_fake)
decode_orient = _define('decode_orient', _decode_orient,
                        lambda o: o.new_empty((o.shape[0], o.shape[2], o.shape[3]), dtype=torch.uint8))
endp_topk = _define('endp_topk', _endp_topk, _endp_topk_fake)


# ------------------------------------------------------------------------------------------------------------- stage ops
def _bev_raster(points: Tensor, tile_offsets: List[int], params: Tensor, H: int, W: int) -> Tensor:
    """points [sum N,4] f32 on the device, tile_offsets B+1 ints, params [B,15] f32 on the HOST (LmRasterParams fields in order:
    quat 4, trans 3, bev_img_offset 2, img_reso 2, local_min_ele, ele_reso, inten_lo, inten_hi) -> u8 HWC tiles [B,H,W,3]."""
    p = params.detach().cpu().float().numpy()
    pars = [_ops.make_raster_params(quat=r[0:4], trans=r[4:7], bev_img_offset=r[7:9], img_reso=r[9:11], local_min_ele=r[11],
                                    ele_reso=r[12], inten_lo=r[13], inten_hi=r[14]) for r in p]
    return _ops.bev_raster_batch(points, tile_offsets, pars, H, W, u8_only=True)


bev_raster = _define('bev_raster', _bev_raster,
                     lambda points, tile_offsets, params, H, W: points.new_empty((len(tile_offsets) - 1, H, W, 3), dtype=torch.uint8))


def raster_params_tensor(params):
    """list of LmRasterParams -> the [B,15] float tensor `bev_raster` takes."""
    rows = [[*p.quat, *p.trans, *p.bev_img_offset, *p.img_reso, p.local_min_ele, p.ele_reso, p.inten_lo, p.inten_hi] for p in params]
    return torch.tensor(rows, dtype=torch.float32)


def _fpn_encoder(proj: Tensor, fea_up_out: Tensor, weights: List[Tensor], stage: str) -> Tuple[Tensor, Tensor, Tensor]:
    """FPN encoder stage (pcencoder.FPNEncoder): proj [B,3,H,W] f32 or [B,H,W,3] u8 -> (fea [B,64,H/8,W/8], bi_seg [B,3,H,W],
    endp [B,1,H,W]); fea_up [B,8,H/4,W/4] is written into `fea_up_out` (a channel slice of the head's concat buffer is fine).
    weights = stage_weights(module), stage = stage_name(module)."""
    with _with_weights(stage, weights) as m:
        fea, _, bi_seg, endp = m._forward_impl(proj, fea_up_out)
    return fea, bi_seg, endp


def _fpn_encoder_fake(proj, fea_up_out, weights, stage):
    B, H, W = (proj.shape[0], proj.shape[1], proj.shape[2]) if proj.dtype == torch.uint8 else (proj.shape[0], proj.shape[2], proj.shape[3])
    m = _stage(stage)
    cf = m.out.out_channels if m.out is not None else 256
    return (_fake_act(proj, B, cf, H // 8, W // 8), proj.new_empty((B, 3, H, W), dtype=torch.float32),
            proj.new_empty((B, 1, H, W), dtype=torch.float32))


fpn_encoder = _define('fpn_encoder', _fpn_encoder, _fpn_encoder_fake, mutates=('fea_up_out',))


def _vit_backbone(fea: Tensor, weights: List[Tensor], stage: str) -> Tensor:
    with _with_weights(stage, weights) as m:
        return m._forward_impl(fea)


def _vit_backbone_fake(fea, weights, stage):
    m = _stage(stage)
    c = m.shared_mlp.out_channels if m.is_with_shared_mlp else m.out_c
    return _fake_act(fea, fea.shape[0], c, m.grid * m.patch, m.grid * m.patch)


vit_backbone = _define('vit_backbone', _vit_backbone, _vit_backbone_fake)


def _colprop_head(x: Tensor, col: Tensor, weights: List[Tensor], stage: str) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """Column-proposal head stage (heads.ColumnProposal2): x [B,8,144,144], col [B,16,288,288] whose channels 8..15 hold fea_up
    (channels 0..7 are filled here) -> (proposal_conf, ext2, cls2, offset2, orient logits)."""
    with _with_weights(stage, weights) as m:
        o = m._forward_impl(x, None, None, col=col)
    return o['proposal_conf'], o['ext2'], o['cls2'], o['offset2'], o['orient']


def _colprop_head_fake(x, col, weights, stage):
    m = _stage(stage)
    B, R, P = x.shape[0], x.shape[2], m.num_prop
    return (x.new_empty((B, P, 2)), x.new_empty((B, P, R, 3)), x.new_empty((B, P, R, 10)), x.new_empty((B, P, R, 10)),
            _fake_act(x, B, m.num_orients, x.shape[2], x.shape[3]))


colprop_head = _define('colprop_head', _colprop_head, _colprop_head_fake, mutates=('col',))


# ------------------------------------------------------------------------------------------------------------- host ops
def _endp_cluster(topk_idx: Tensor, crop_w: int, clip: int, k0: int, k_max: int) -> Tensor:
    from . import hostpost
    pts, _ = hostpost.cluster_endpoints(topk_idx.numpy(), crop_w=crop_w, clip=clip, k0=k0, k_max=k_max)
    return torch.from_numpy(pts.astype(np.int32).reshape(-1, 2))


def _endp_cluster_fake(topk_idx, crop_w, clip, k0, k_max):
    n = torch.library.get_ctx().new_dynamic_size()
    return topk_idx.new_empty((n, 2), dtype=torch.int32)


endp_cluster = _define('endp_cluster', _endp_cluster, _endp_cluster_fake, device='cpu')


def _polyline_assemble(prop_conf: Tensor, prop_v_ext: Tensor, cls_offset: Tensor, bi_seg_rows: Tensor, endp_hw: Tensor,
                       obj_thre: float) -> Tuple[Tensor, Tensor]:
    from . import hostpost
    lanes, kept = hostpost.assemble_polylines(prop_conf.numpy(), prop_v_ext.numpy(), cls_offset.numpy(), bi_seg_rows.numpy(),
                                              endp_hw.numpy(), obj_thre)
    return torch.from_numpy(lanes), torch.from_numpy(np.ascontiguousarray(kept, dtype=np.int32).reshape(-1, 2))


def _polyline_assemble_fake(prop_conf, prop_v_ext, cls_offset, bi_seg_rows, endp_hw, obj_thre):
    n = torch.library.get_ctx().new_dynamic_size()
    P, R = prop_v_ext.shape
    return prop_conf.new_empty((P, R, 2), dtype=torch.float64), endp_hw.new_empty((n, 2), dtype=torch.int32)


polyline_assemble = _define('polyline_assemble', _polyline_assemble, _polyline_assemble_fake, device='cpu')

OP_NAMES = ['conv2d_mfma', 'conv3x3_winograd44', 'stem_conv7x7', 'maxpool3x3s2', 'gn_stats', 'gn_relu_upsample',
            'upsample_bilinear', 'layernorm_rows', 'attention', 'linear_mfma', 'tile_ingest', 'decode_proposals', 'decode_semantic',
            'decode_orient', 'endp_topk', 'bev_raster', 'fpn_encoder', 'vit_backbone', 'colprop_head', 'endp_cluster', 'polyline_assemble']
