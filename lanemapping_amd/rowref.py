"""K-Lane "RowRef" head behind the reference's HEADS registry name ``RowSharNotReducRef`` (BASELINE config 4).

Drop-in for baseline/models/heads/row_shared_not_reduc_ref.py: same kwargs (:88-103), same state-dict keys
(`ext_c / cls_c / ext2_c / cls2_c`, `to_token.1`, `tr_lane_correlator.{0,1,2}`), same output dictionary
(`ext_c, cls_c, ext2_c, cls2_c` soft-maxed, :170-246) and `get_exist_coor_endp_dict` -> {'conf','cls'} (:334-363).

`emb_c`: the reference creates them as `nn.Parameter(torch.randn(dim_token)).cuda()` (:140), i.e. on a real GPU they are
plain tensors that are NOT in checkpoints and are re-drawn from the seeded RNG at construction (SURVEY §8c caveat 2).
Here they are non-persistent buffers drawn the same way; `set_lane_embeddings` lets a caller supply captured values.

GPU plan: the 24 first-stage Conv1d+BN1d of a stage are ONE GEMM [B*144, 1152] x [1152, 12288] straight from the NHWC
feature map (weight columns permuted from (c w) to (w c) at pack time); second convs are per-lane GEMMs into column
slices of the ext / cls buffers; softmax, lane selection, window gather / scatter and decode are small kernels
(csrc/rowref.hip); the lane-token transformer reuses the ViT kernels.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops, hostpost
from ._lib import lib, check
from .backbone import _Transformer, pack_transformer, transformer_forward
from .registry import HEADS
from .packing import PackedModule


def _stack(cin, hidden, cout):
    return nn.Sequential(nn.Conv1d(cin, hidden, 1), nn.BatchNorm1d(hidden), nn.Conv1d(hidden, cout, 1), nn.Identity())


class _Last(dict):
    """Tensors of the last forward; 'selected' (the [B, L] boolean lane-selection mask the reference computes on the host) is only
    materialised when somebody asks for it - the forward itself never reads it back."""

    def __missing__(self, key):
        if key == 'selected':
            return self['valid'].cpu().numpy() != 0
        raise KeyError(key)


@HEADS.register_module
class RowSharNotReducRef(PackedModule):
    def __init__(self, dim_feat=8, row_size=144, dim_shared=512, lambda_cls=1., thr_ext=0.3, off_grid=2, dim_token=1024,
                 tr_depth=1, tr_heads=16, tr_dim_head=64, tr_mlp_dim=2048, tr_dropout=0., tr_emb_dropout=0.,
                 is_reuse_same_network=False, cfg=None):
        super().__init__()
        if dim_feat != 8 or off_grid != 2 or is_reuse_same_network:
            raise NotImplementedError('hot path covers dim_feat=8, off_grid=2, separate 2nd-stage networks (config 4)')
        self.cfg = cfg
        self.num_cls = cfg.number_lanes
        self.row_size, self.dim_shared, self.dim_token = row_size, dim_shared, dim_token
        self.thr_ext, self.off_grid = thr_ext, off_grid
        self.is_reuse_same_network = False
        cin = dim_feat * row_size
        for c in range(self.num_cls):
            setattr(self, f'ext_{c}', _stack(cin, dim_shared, 2))
            setattr(self, f'cls_{c}', _stack(cin, dim_shared, row_size))
        in_tok = (2 * off_grid + 1) * row_size * dim_feat
        self.to_token = nn.Sequential(nn.Identity(), nn.Linear(in_tok, dim_token))
        for c in range(self.num_cls):
            self.register_buffer(f'emb_{c}', torch.randn(dim_token), persistent=False)
        self.tr_lane_correlator = nn.Sequential(_Transformer(dim_token, tr_depth, tr_heads, tr_dim_head, tr_mlp_dim),
                                                nn.LayerNorm(dim_token), nn.Linear(dim_token, in_tok), nn.Identity())
        for c in range(self.num_cls):
            setattr(self, f'ext2_{c}', _stack(cin, dim_shared, 2))
            setattr(self, f'cls2_{c}', _stack(cin, dim_shared, row_size))

    def set_lane_embeddings(self, values):
        for c, v in enumerate(values):
            getattr(self, f'emb_{c}').copy_(torch.as_tensor(v))

    # ------------------------------------------------------------------------------------------ packing
    def _pack_stage(self, P, tag, ext_name, cls_name):
        W, S, Bs = [], [], []
        C, R = 8, self.row_size
        for c in range(self.num_cls):
            for name in (ext_name, cls_name):
                st = getattr(self, f'{name}_{c}')
                w = st[0].weight[:, :, 0]                                        # [512, (c w)]
                W.append(w.reshape(-1, C, R).permute(0, 2, 1).reshape(-1, C * R))   # -> (w c) = NHWC row order
                sc, sh = ops.fold_bn(st[1], st[0].bias)
                S.append(sc)
                Bs.append(sh)
                P[f'{tag}.{name}{c}.w2'] = ops.pack_mfma(st[2].weight[:, :, 0])
                P[f'{tag}.{name}{c}.b2'] = st[2].bias.float().contiguous()
        P[tag + '.w1'] = ops.pack_mfma(torch.cat(W, dim=0))
        P[tag + '.s1'], P[tag + '.b1'] = torch.cat(S).contiguous(), torch.cat(Bs).contiguous()

    def _pack(self):
        P = {}
        self._pack_stage(P, 's1', 'ext', 'cls')
        self._pack_stage(P, 's2', 'ext2', 'cls2')
        P['tok.w'] = ops.pack_mfma(self.to_token[1].weight)
        P['tok.b'] = self.to_token[1].bias.float().contiguous()
        P['emb'] = torch.stack([getattr(self, f'emb_{c}') for c in range(self.num_cls)]).float().contiguous()
        pack_transformer(self.tr_lane_correlator[0].layers, P, 'T')
        ln, lin = self.tr_lane_correlator[1], self.tr_lane_correlator[2]
        P['ln.g'], P['ln.b'] = ln.weight.float().contiguous(), ln.bias.float().contiguous()
        P['exp.w'] = ops.pack_mfma(lin.weight)
        P['exp.b'] = lin.bias.float().contiguous()
        return P

    # ------------------------------------------------------------------------------------------ forward
    def _stage(self, P, tag, ext_name, cls_name, x):
        """x [B,8,H,W] NHWC-stored -> ext [B,H,L,2], cls [B,H,L,W] probabilities."""
        x, ld = ops.as_nhwc(x)
        B, C, H, W = x.shape
        assert ld == C
        L, D = self.num_cls, self.dim_shared
        A = x.permute(0, 2, 3, 1).reshape(B * H, W * C)
        hid = ops.linear_mfma(A, P[tag + '.w1'], 2 * L * D, scale=P[tag + '.s1'], shift=P[tag + '.b1'])
        ext = torch.empty((B * H, L * 2), device=x.device, dtype=torch.float32)
        cls = torch.empty((B * H, L * W), device=x.device, dtype=torch.float32)
        for c in range(L):
            ops.linear_mfma(hid[:, (2 * c) * D:(2 * c + 1) * D], P[f'{tag}.{ext_name}{c}.w2'], 2,
                            shift=P[f'{tag}.{ext_name}{c}.b2'], out=ext[:, 2 * c:2 * c + 2])
            ops.linear_mfma(hid[:, (2 * c + 1) * D:(2 * c + 2) * D], P[f'{tag}.{cls_name}{c}.w2'], W,
                            shift=P[f'{tag}.{cls_name}{c}.b2'], out=cls[:, c * W:(c + 1) * W])
        check(lib().lm_softmax_rows(ops._stream(), ops._ptr(ext), B * H * L, 2))
        check(lib().lm_softmax_rows(ops._stream(), ops._ptr(cls), B * H * L, W))
        return ext.view(B, H, L, 2), cls.view(B, H, L, W)

    def forward(self, x):
        P = self.packed()
        x, _ = ops.as_nhwc(x)
        B, C, H, W = x.shape
        L = self.num_cls
        self.b_size = B
        dev = x.device
        ext1, cls1 = self._stage(P, 's1', 'ext', 'cls', x)
        # lane selection (:199-204) stays on the device: tokens live on the fixed grid t = b * L + lane, `valid` flags the selected ones
        # (round 2 read the mask on the host here, which serialised the streams and ruled out graph capture)
        mean = torch.empty((B, L), device=dev, dtype=torch.float32)
        valid = torch.empty((B, L), device=dev, dtype=torch.int32)
        corr = torch.empty((B, L, H), device=dev, dtype=torch.int32)
        check(lib().lm_rowref_select(ops._stream(), ops._ptr(ext1), ops._ptr(cls1), ops._ptr(mean), ops._ptr(valid), float(self.thr_ext),
                                     ops._ptr(corr), B, H, W, L))
        tok = torch.empty((B * L, C * H * 5), device=dev, dtype=torch.float32)
        check(lib().lm_rowref_gather(ops._stream(), ops._ptr(x), ops._ptr(corr), ops._ptr(tok), B, H, W, L))
        t = ops.linear_mfma(tok, P['tok.w'], self.dim_token, shift=P['tok.b'], res=P['emb'], res_rows=L)      # + lane embedding (row t: lane t % L)
        # attention only among the selected lanes of one tile: keys = the flagged tokens of the tile, in lane order
        t = transformer_forward(self.tr_lane_correlator[0].layers, P, 'T', t, B, L, valid=valid)
        t = ops.layernorm(t.contiguous(), P['ln.g'], P['ln.b'], self.tr_lane_correlator[1].eps)
        t = ops.linear_mfma(t, P['exp.w'], C * H * 5, shift=P['exp.b'])
        x2 = ops.new_act(B, C, H, W, dev)
        check(lib().lm_rowref_scatter(ops._stream(), ops._ptr(x), ops._ptr(t), ops._ptr(corr), ops._ptr(valid), ops._ptr(x2), B, H, W, L))
        ext2, cls2 = self._stage(P, 's2', 'ext2', 'cls2', x2)
        out = {}
        for c in range(L):
            out[f'ext_{c}'], out[f'cls_{c}'] = ext1[:, :, c, :], cls1[:, :, c, :]
            out[f'ext2_{c}'], out[f'cls2_{c}'] = ext2[:, :, c, :], cls2[:, :, c, :]
        self._last = _Last({'ext2': ext2, 'cls2': cls2, 'refined': x2, 'valid': valid})
        return out

    # ------------------------------------------------------------------------------------------ decode / lines
    def get_exist_coor_endp_dict(self, out, is_get_1_stage_result=False):
        if is_get_1_stage_result:
            raise NotImplementedError('first-stage maps are a debugging aid of the reference; not on the hot path')
        ext2, cls2 = self._last['ext2'], self._last['cls2']
        assert out['ext2_0'].data_ptr() == ext2.data_ptr(), 'decode expects the dictionary of the last forward()'
        B, H, L, W = cls2.shape
        dev = cls2.device
        conf = torch.empty((B, H, W), device=dev, dtype=torch.uint8)
        cmap = torch.empty((B, L + 1, H, W), device=dev, dtype=torch.uint8)
        col = torch.empty((B, L, H), device=dev, dtype=torch.int32)
        check(lib().lm_rowref_decode(ops._stream(), ops._ptr(ext2), ops._ptr(cls2), ops._ptr(conf), ops._ptr(cmap), ops._ptr(col), B, H, W, L))
        self._col_idx = col
        return {'conf': conf.cpu().to(torch.float64), 'cls': cmap.cpu().to(torch.float64)}

    def decode_columns(self, out=None):
        """Device part of the decode the tile pipeline needs: col_idx [B,L,H] i32 (argmax column, -1 = row absent), no dense maps."""
        ext2, cls2 = self._last['ext2'], self._last['cls2']
        B, H, L, W = cls2.shape
        col = torch.empty((B, L, H), device=cls2.device, dtype=torch.int32)
        check(lib().lm_rowref_decode(ops._stream(), ops._ptr(ext2), ops._ptr(cls2), None, None, ops._ptr(col), B, H, W, L))
        return col

    @staticmethod
    def lines_from_columns(col, row_size=144):
        """One tile, host: col [L,H] i32 -> traced lines [L,H] f64 column px (reference :487-516: a pixel claimed by a lower lane
        index wins, col / row_size * 1152 + 4, then smooth_cls_line_per_batch with constant orientation / no segmentation map)."""
        L, H = col.shape
        lines = np.zeros((L, H)) - 1.0
        for c in range(L):
            ok = col[c] >= 0
            if c:
                ok &= ~(col[:c] == col[c][None, :]).any(axis=0)         # (h, col) already claimed by a lower lane index
            lines[c, ok] = col[c, ok] / row_size * 1152. + 4
        return hostpost.trace_lines(lines)

    def predict_lines(self):
        """Label-free part of get_lane_map_numpy_with_label (:487-516): per tile [L,144] column px after
        smooth_cls_line_per_batch (constant orientation, no segmentation confidence)."""
        col = self._col_idx.cpu().numpy()
        return [self.lines_from_columns(col[b], self.row_size) for b in range(col.shape[0])]

    def get_lane_map_numpy_with_label(self, output, data, is_flip=True, is_img=False, is_get_1_stage_result=True, is_gt_avai=False):
        """Only the label-free outputs of the reference method (it otherwise needs GT tensors): `cls_offset_smooth`.
        Accepts the `is_gt_avai` keyword Detector1stage passes (the reference method does not: SURVEY quirk C10)."""
        return {'coor_label': [], 'cls_offset_smooth': self.predict_lines()}
