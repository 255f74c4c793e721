"""Column-proposal head behind the reference's HEADS registry name ``ColumnProposal2``.

Drop-in for baseline/models/heads/polyline_fpn_vit_vertex_2.py: same constructor kwargs (:66-99), same
state-dict keys (3.95 M parameters, of which only ~0.10 M are live at inference — SURVEY F9; the dead ones
are kept so checkpoints load strictly), same method names and output dictionaries:

  forward(x, x_up, x_endp)                 :309-435  -> proposal_conf, ext2, cls2, offset2, orient
  get_exist_coor_endp_dict(out)            :602-759  -> prop_conf, prop_v_ext, prop_cls_conf, endp, orient, bi_seg,
                                                        semantic_seg, cls_offset
  get_lane_map_numpy_with_label(...)       :761-886  -> lane_maps {coor_label, cls_offset_smooth, endp_by_cls, semantic_line}
  get_lane_map_on_source_image(...)        :926-1083 -> pred_smooth_lane_vertex (vertex packing only, no cv2 overlays)

Not supported (raise): column_att / column_transformer_decoder branches, endp_mode == 'endpoint',
view_detail=True (the reference itself raises NameError there).  `prop_bi_seg`
([B,72,1,1152,80], unused downstream) and the dead `endpoint` map are not produced.
"""
import numpy as np
import torch
import torch.nn as nn

from . import ops, decode, hostpost
from .backbone import _Transformer, _FeedForward, pack_transformer
from .registry import HEADS
from .packing import PackedModule


class _ConvPool2d(nn.Module):
    """Parameter container of the (dead at inference) `generate_line_proposal` stack (:48-61)."""

    def __init__(self, cin, hidden, cout):
        super().__init__()
        layers = [nn.Conv2d(cin, cin, (5, 3), padding=(2, 1))]
        for a, b in zip([cin] + hidden, hidden + [cout]):
            layers.append(nn.Sequential(nn.ReLU(inplace=True), nn.BatchNorm2d(a), nn.Conv2d(a, b, 3, 2, 1)))
        self.layers = nn.ModuleList(layers)


def _conv1d_stack(cin, hidden, cout):
    return nn.Sequential(nn.Conv1d(cin, hidden, 1), nn.BatchNorm1d(hidden), nn.Conv1d(hidden, cout, 1), nn.Identity())


@HEADS.register_module
class ColumnProposal2(PackedModule):
    def __init__(self, dim_feat=8, row_size=144, dim_shared=512, num_prop=72, prop_width=2, prop_half_buff=4,
                 dim_token=1024, tr_depth=1, tr_heads=16, tr_dim_head=64, tr_mlp_dim=2048, tr_dropout=0.,
                 tr_emb_dropout=0., row_dim_token=64, row_tr_depth=1, row_tr_heads=10, row_tr_dim_head=12,
                 row_tr_mlp_dim=128, row_tr_dropout=0., row_tr_emb_dropout=0., endp_mode='Regr', cls_exp=False,
                 ext_w=1., ext_smooth_w=1., lambda_cls=1., mean_loss_w=0., cls_smooth_loss_w=0., orient_w=1.,
                 endp_loss_w=1., offset_w=1., freeze_endp=False, freeze_ori=False, cfg=None):
        super().__init__()
        self.cfg = cfg
        self.flip_label = cfg.flip_label
        self.num_cls = cfg.number_lanes
        self.num_orients = cfg.number_orients
        self.num_prop, self.prop_width, self.prop_half_buff = num_prop, prop_width, prop_half_buff
        self.row_size, self.endp_mode = row_size, endp_mode
        self.N_s = prop_width
        self.prop_fea_width = prop_width + 2 * prop_half_buff
        self.dim_shared = dim_shared
        hd = dim_feat * 2
        # ---- dead-at-inference parameters (column_att branch), kept for strict checkpoint loading ----
        self.reg_ffn = _FeedForward(dim_feat, dim_feat * 4)
        hidden = {72: [], 36: [2 * dim_feat], 18: [2 * dim_feat, 4 * dim_feat]}.get(num_prop)
        if hidden is not None:
            self.generate_line_proposal = nn.Sequential(_ConvPool2d(dim_feat, hidden, dim_feat * 2 ** (len(hidden) + 1)))
        in_tok = num_prop * dim_feat * prop_width
        self.to_token = nn.Sequential(nn.Identity(), nn.Linear(in_tok, dim_token))
        for i in range(num_prop):
            setattr(self, f'emb_{i}', nn.Parameter(torch.randn(dim_token)))
        self.tr_lane_correlator = nn.Sequential(_Transformer(dim_token, tr_depth, tr_heads, tr_dim_head, tr_mlp_dim),
                                                nn.LayerNorm(dim_token))
        self.line_expand = nn.Sequential(nn.Linear(dim_token, in_tok), nn.Identity())
        self.head_upsample_layers = nn.Sequential(nn.Conv2d(hd, dim_feat, (5, 3), 1, (2, 1)), nn.BatchNorm2d(dim_feat),
                                                  nn.Conv2d(dim_feat, dim_feat, 3, 1, 1), nn.BatchNorm2d(dim_feat))
        self.endpoint = nn.Sequential(nn.Conv2d(hd + 1, dim_feat // 2, 3, 1, 1), nn.ReLU(inplace=True),
                                      nn.BatchNorm2d(dim_feat // 2), nn.Conv2d(dim_feat // 2, 1, 3, 1, 1))
        # ---- live parameters ----
        self.head_common_layers = nn.Sequential(nn.Conv2d(hd, hd, 3, 1, 1), nn.BatchNorm2d(hd),
                                                nn.Conv2d(hd, hd, 3, 2, 1), nn.BatchNorm2d(hd))
        self.proposal_confidence = nn.Sequential(nn.Identity(), nn.Linear(hd * self.prop_fea_width * row_size, 2))
        self.ext2 = _conv1d_stack(hd * self.prop_fea_width, dim_shared, 3)
        self.cls2 = _conv1d_stack(hd * self.prop_fea_width, dim_shared, self.prop_fea_width)
        self.offset2 = _conv1d_stack(hd * self.prop_fea_width, dim_shared, self.prop_fea_width)
        self.orient = nn.Sequential(nn.Conv2d(hd, hd // 2, 3, 1, 1), nn.BatchNorm2d(hd // 2),
                                    nn.Conv2d(hd // 2, self.num_orients, 3, 1, 1))
        self.bi_seg_proposal = nn.Conv2d(hd, 1, 1)

    # -------------------------------------------------------------------------------- packing
    def _pack(self):
        P = {}
        hc = self.head_common_layers
        P['hc0.w'] = ops.pack_small(hc[0].weight)
        P['hc0.s'], P['hc0.b'] = ops.fold_bn(hc[1], hc[0].bias)
        P['hc2.w'] = ops.pack_small(hc[2].weight)
        P['hc2.s'], P['hc2.b'] = ops.fold_bn(hc[3], hc[2].bias)
        P['or0.w'] = ops.pack_small(self.orient[0].weight)
        P['or0.s'], P['or0.b'] = ops.fold_bn(self.orient[1], self.orient[0].bias)
        P['or2.w'] = ops.pack_small(self.orient[2].weight)
        P['or2.b'] = self.orient[2].bias.float().contiguous()
        P['seg.w'] = ops.pack_small(self.bi_seg_proposal.weight)
        P['seg.b'] = self.bi_seg_proposal.bias.float().contiguous()
        P['seg.bias_value'] = float(self.bi_seg_proposal.bias.item())
        stacks = (self.ext2, self.cls2, self.offset2)
        P['w1'] = ops.pack_mfma(torch.cat([s[0].weight[:, :, 0] for s in stacks], dim=0))
        sc, sh = zip(*[ops.fold_bn(s[1], s[0].bias) for s in stacks])
        P['s1'], P['b1'] = torch.cat(sc).contiguous(), torch.cat(sh).contiguous()
        P['w2'] = torch.cat([s[2].weight[:, :, 0] for s in stacks], dim=0).float().contiguous()
        P['b2'] = torch.cat([s[2].bias for s in stacks]).float().contiguous()
        lin = self.proposal_confidence[1]
        cw = lin.in_features // self.row_size
        P['conf.w'] = lin.weight.reshape(2, cw, self.row_size).permute(0, 2, 1).reshape(2, -1).float().contiguous()
        P['conf.b'] = lin.bias.float().contiguous()
        return P

    # -------------------------------------------------------------------------------- forward
    def forward(self, x, x_up, x_endp=None, col=None):
        """x [B,8,144,144], x_up [B,8,288,288] -> raw head outputs (live sub-graph).
        `col`: optional pre-assembled [B,16,288,288] buffer whose channels 8..15 already hold x_up.
        Goes through the dispatcher: torch.ops.lanemap_hip.colprop_head (torch_ops.py)."""
        from . import torch_ops
        if col is None:
            col = ops.new_act(x.shape[0], 16, x_up.shape[2], x_up.shape[3], x.device)
            col[:, 8:16].copy_(x_up)
        self.b_size = x.shape[0]
        conf, ext2, cls2, off2, orient = torch_ops.colprop_head(x, col, torch_ops.stage_weights(self), torch_ops.stage_name(self))
        return {'proposal_conf': conf, 'ext2': ext2, 'cls2': cls2, 'offset2': off2, 'orient': orient}

    def _forward_impl(self, x, x_up, x_endp=None, col=None):
        cfg = self.cfg
        if cfg.column_att or cfg.column_transformer_decoder:
            raise NotImplementedError('column_att / column_transformer_decoder are off in every BASELINE config')
        if not cfg.spatial_att:
            raise NotImplementedError('spatial_att=False is not on the hot path')
        if self.prop_fea_width != 10 or self.dim_shared * 3 > 320:
            raise NotImplementedError('hot path covers prop_fea_width == 10 and dim_shared <= 106')
        P = self.packed()
        B, _, h, w = x.shape
        self.b_size = B
        hd = 16
        if col is None:
            col = ops.new_act(B, hd, x_up.shape[2], x_up.shape[3], x.device)
            col[:, 8:16].copy_(x_up)
        ops.upsample_nhwc(x, col.shape[2:], out=col[:, 0:8])                                   # :359
        r = ops.conv_small(col, P['hc0.w'], hd, 3, 3, 1, 1, scale=P['hc0.s'], shift=P['hc0.b'])
        row = ops.conv_small(r, P['hc2.w'], hd, 3, 3, 2, 1, scale=P['hc2.s'], shift=P['hc2.b'])   # :376
        o = ops.conv_small(row, P['or0.w'], hd // 2, 3, 3, 1, 1, scale=P['or0.s'], shift=P['or0.b'])
        orient = ops.conv_small(o, P['or2.w'], self.num_orients, 3, 3, 1, 1, shift=P['or2.b'])      # :380
        seg = ops.conv_small(col, P['seg.w'], 1, shift=P['seg.b'], pre_relu=True)                   # :400 (once)
        tok = ops.head_tokens(seg, row, self.num_prop, self.prop_width, self.prop_half_buff, P['seg.bias_value'])
        D = self.dim_shared
        hid = torch.empty((tok.shape[0], 320), device=x.device, dtype=torch.float32)
        ops.linear_mfma(tok, P['w1'], 3 * D, scale=P['s1'], shift=P['b1'], out=hid)
        ext2, cls2, off2 = ops.head_stage2(hid, D, P['w2'], P['b2'], B, self.num_prop, h)
        conf = ops.head_proposal_conf(tok, P['conf.w'], P['conf.b'], B, self.num_prop)
        return {'proposal_conf': conf, 'ext2': ext2, 'cls2': cls2, 'offset2': off2, 'orient': orient}

    # -------------------------------------------------------------------------------- decode / assembly
    def decode_compact(self, out):
        """Device decode + endpoint clustering, compact form (what the runner and bench use)."""
        if self.cfg.heads.endp_mode == 'endpoint':
            raise NotImplementedError("endp_mode='endpoint' (dead branch) is not supported")
        return decode.decode_compact(out, self.cfg, self.num_cls, self.prop_width, self.prop_half_buff)

    def get_exist_coor_endp_dict(self, out):
        if getattr(self.cfg, 'view_detail', False):
            raise NotImplementedError('view_detail=True is unsupported (the reference raises NameError there)')
        c = self.decode_compact(out)
        self._compact = c
        return decode.compact_to_reference_dict(c)

    def get_lane_map_numpy_with_label(self, output, data, is_flip=True, is_img=False, is_get_1_stage_result=False,
                                      is_gt_avai=True):
        B = output['prop_conf'].shape[0]
        lane_maps = {'coor_label': [], 'cls_offset_smooth': [], 'endp_by_cls': [], 'semantic_line': []}
        if is_gt_avai:
            lane_maps['coor_label'] = [data['lc_coor_raw'][b].cpu().numpy() for b in range(B)]
        pc = output['prop_conf'].float().cpu().numpy()
        ve = output['prop_v_ext'].float().cpu().numpy()
        co = output['cls_offset'].double().cpu().numpy()
        comp = getattr(self, '_compact', None)
        if comp is not None and comp.get('bi_seg_rows') is not None and comp['bi_seg'] is output.get('bi_seg'):
            rows = comp['bi_seg_rows'].cpu().numpy()
        else:
            rows = output['bi_seg'][:, 3::8, :].float().cpu().numpy()
        for b in range(B):
            e = output['endp'][b]
            ep = np.stack(np.nonzero(e.cpu().numpy() if torch.is_tensor(e) else e), axis=1).astype(np.int32)
            lanes, kept = hostpost.assemble_polylines(pc[b], ve[b], co[b], rows[b], ep, self.cfg.proposal_obj_thre)
            emap = np.zeros((self.row_size * 8, self.row_size * 8), dtype=np.float32)
            emap[kept[:, 0], kept[:, 1]] = 1.0
            lane_maps['cls_offset_smooth'].append(lanes)
            lane_maps['endp_by_cls'].append(emap)
            lane_maps['semantic_line'].append(hostpost.raster_semantic_map(lanes))
        return lane_maps

    def get_lane_map_on_source_image(self, output, data, is_img=True):
        """Only the vertex packing (row = 3 + 8 i, col, semantic) of :997-1000,1056; overlays are out of scope."""
        packed = []
        for lanes in output['lane_maps']['cls_offset_smooth']:
            v = np.zeros((lanes.shape[0], self.row_size, 3))
            v[:, :, 0] = np.arange(3, self.row_size * 8, 8)
            v[:, :, 1:] = lanes
            packed.append(v)
        return {'pred_smooth_lane_vertex': packed}
