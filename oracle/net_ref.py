"""Oracle (test infrastructure): torch-CPU fp32 restatement of the network forward.

Functional style over a flat ``state_dict`` that uses the reference's parameter names, so
the same synthetic weights drive the reference (golden generation), this oracle and the
HIP product.  Rows of SURVEY.md §8a:

  a3  fpn_forward   <- baseline/models/pcencoder/postprojector.py:563-655 (blocks :299-338,
                       ctor :417-515, _make_layer :517-539)
  a4  vit_forward   <- baseline/models/backbone/vitsegnet.py:194-214 (Attention :41-68,
                       FeedForward :28-39, Transformer :70-83)
  a5  head_forward  <- baseline/models/heads/polyline_fpn_vit_vertex_2.py:309-435
"""
import torch
import torch.nn.functional as F

BN_EPS = 1e-5


def _bn(x, sd, p):
    return F.batch_norm(x, sd[p + '.running_mean'], sd[p + '.running_var'],
                        sd[p + '.weight'], sd[p + '.bias'], False, 0.0, BN_EPS)


def _up(x, h, w):
    # postprojector.py:541-542 — bilinear, align_corners=True (used even when size is unchanged)
    return F.interpolate(x, size=(h, w), mode='bilinear', align_corners=True)


def _basic_block(x, sd, p, stride, dilation):
    # postprojector.py:322-338
    y = F.conv2d(x, sd[p + '.conv1.weight'], None, stride, dilation, dilation)
    y = F.relu(_bn(y, sd, p + '.bn1'))
    y = F.conv2d(y, sd[p + '.conv2.weight'], None, 1, dilation, dilation)
    y = _bn(y, sd, p + '.bn2')
    if (p + '.downsample.0.weight') in sd:
        x = _bn(F.conv2d(x, sd[p + '.downsample.0.weight'], None, stride), sd, p + '.downsample.1')
    return F.relu(y + x)


def fpn_forward(sd, x, p='pcencoder.fpn'):
    """x [B,3,H,W] -> (fea [B,64,H/8,W/8], fea_up [B,8,H/4,W/4], bi_seg [B,3,H,W], endp [B,1,H,W]).

    Config in_channels=[64,128,256,-1], replace_stride_with_dilation=[F,T,F]
    (configs/Proj_polyline_fpn_vit_vertex_2.py:29-36)."""
    H, W = x.shape[2:]
    g = lambda k: sd[p + '.' + k]
    # stem  :566-567
    c1 = F.relu(_bn(F.conv2d(x, g('conv1.weight'), None, 2, 3), sd, p + '.bn1'))
    c1 = F.max_pool2d(c1, 3, 2, 1)
    # layer1: 3 blocks, stride 1  :568
    c2 = c1
    for i in range(3):
        c2 = _basic_block(c2, sd, f'{p}.layer1.{i}', 1, 1)
    # layer2: 4 blocks, first stride 2  :569
    c3 = c2
    for i in range(4):
        c3 = _basic_block(c3, sd, f'{p}.layer2.{i}', 2 if i == 0 else 1, 1)
    # layer3: 6 blocks; stride replaced by dilation 2, first block keeps dilation 1  :520-537,572
    c4 = c3
    for i in range(6):
        c4 = _basic_block(c4, sd, f'{p}.layer3.{i}', 1, 1 if i == 0 else 2)
    fea = F.conv2d(c4, g('out.weight'))                                             # :573-574
    # top-down  :591-593
    p4 = F.conv2d(c4, g('toplayer.weight'), g('toplayer.bias'))
    l1 = F.conv2d(c3, g('latlayer1.weight'), g('latlayer1.bias'))
    p3 = _up(p4, l1.shape[2], l1.shape[3]) + l1
    l2 = F.conv2d(c2, g('latlayer2.weight'), g('latlayer2.bias'))
    p2 = _up(p3, l2.shape[2], l2.shape[3]) + l2
    # smooth  :597-599
    p4 = F.conv2d(p4, g('smooth1.weight'), g('smooth1.bias'), 1, 1)
    p3 = F.conv2d(p3, g('smooth2.weight'), g('smooth2.bias'), 1, 1)
    p2 = F.conv2d(p2, g('smooth3.weight'), g('smooth3.bias'), 1, 1)
    h, w = p2.shape[2:]

    def gn(t, name):   # GroupNorm(C groups == C channels), eps 1e-5  :512-515
        return F.relu(F.group_norm(t, t.shape[1], g(name + '.weight'), g(name + '.bias'), 1e-5))

    def branch(conv_a, gn_a, conv_b, gn_b):   # :615-621 / :641-647
        s4 = _up(gn(F.conv2d(p4, g(conv_a + '.weight'), g(conv_a + '.bias'), 1, 1), gn_a), h, w)
        s4 = _up(gn(F.conv2d(s4, g(conv_b + '.weight'), g(conv_b + '.bias'), 1, 1), gn_b), h, w)
        s3 = _up(gn(F.conv2d(p3, g(conv_b + '.weight'), g(conv_b + '.bias'), 1, 1), gn_b), h, w)
        s2 = gn(F.conv2d(p2, g(conv_b + '.weight'), g(conv_b + '.bias'), 1, 1), gn_b)
        return s2 + s3 + s4

    sa = branch('conv2', 'gn12', 'semantic_branch', 'gn11')
    fea_up = F.conv2d(sa, g('feature_layer.weight'), g('feature_layer.bias'))      # :628
    bi_seg = _up(F.conv2d(F.relu(fea_up), g('output_layer_binary_seg.weight'),
                          g('output_layer_binary_seg.bias')), H, W)                 # :631
    sb = branch('conv3', 'gn22', 'semantic_branch2', 'gn21')
    endp = _up(F.conv2d(sb, g('output_layer_endp.weight'), g('output_layer_endp.bias')), H, W)  # :651
    return fea, fea_up, bi_seg, endp


def vit_forward(sd, x, p='backbone', depth=3, heads=16, dim_head=64, patch=8, out_c=8):
    """x [B,64,144,144] -> [B,8,144,144]  (vitsegnet.py:194-214, is_with_shared_mlp=False)."""
    B, C, H, W = x.shape
    gh, gw = H // patch, W // patch
    # 'b c (h p1) (w p2) -> b (h w) (p1 p2 c)'   :164
    t = x.reshape(B, C, gh, patch, gw, patch).permute(0, 2, 4, 3, 5, 1).reshape(B, gh * gw, patch * patch * C)
    t = F.linear(t, sd[p + '.to_patch_embedding.1.weight'], sd[p + '.to_patch_embedding.1.bias'])
    t = t + sd[p + '.pos_embedding'][:, :gh * gw]                                   # :203
    dim = t.shape[-1]
    scale = dim_head ** -0.5
    for l in range(depth):
        a = f'{p}.transformer.layers.{l}.0'
        f = f'{p}.transformer.layers.{l}.1'
        y = F.layer_norm(t, (dim,), sd[a + '.norm.weight'], sd[a + '.norm.bias'], 1e-5)
        qkv = F.linear(y, sd[a + '.fn.to_qkv.weight'])                              # no bias :51
        q, k, v = [z.reshape(B, -1, heads, dim_head).transpose(1, 2) for z in qkv.chunk(3, dim=-1)]
        dots = torch.matmul(q, k.transpose(-1, -2)) * scale                         # :62
        o = torch.matmul(dots.softmax(dim=-1), v)                                   # :64-66
        o = o.transpose(1, 2).reshape(B, -1, heads * dim_head)
        t = F.linear(o, sd[a + '.fn.to_out.0.weight'], sd[a + '.fn.to_out.0.bias']) + t   # :81
        y = F.layer_norm(t, (dim,), sd[f + '.norm.weight'], sd[f + '.norm.bias'], 1e-5)
        y = F.gelu(F.linear(y, sd[f + '.fn.net.0.weight'], sd[f + '.fn.net.0.bias']))      # exact erf GELU
        t = F.linear(y, sd[f + '.fn.net.3.weight'], sd[f + '.fn.net.3.bias']) + t          # :82
    # 'b (h w) (p1 p2 c) -> b c (h p1) (w p2)'   :180
    t = t.reshape(B, gh, gw, patch, patch, out_c).permute(0, 5, 1, 3, 2, 4).reshape(B, out_c, H, W)
    return t


def head_forward(sd, x, x_up, p='heads', num_prop=72, prop_width=2, half_buff=4, want_prop_bi_seg=False):
    """ColumnProposal2.forward, live sub-graph only (column_att=False, spatial_att=True).

    x [B,8,144,144], x_up [B,8,288,288] -> dict(proposal_conf [B,72,2], ext2 [B,72,144,3],
    cls2 [B,72,144,10], offset2 [B,72,144,10], orient [B,11,144,144]).
    The dead `endpoint` branch (:371-373) is not evaluated."""
    B, _, h, w = x.shape
    fw = prop_width + 2 * half_buff                                                 # :127 -> 10
    col = torch.cat([_up(x, x_up.shape[2], x_up.shape[3]), x_up], dim=1)           # :359 [B,16,288,288]
    # head_common_layers: conv3x3 -> BN -> conv3x3 s2 -> BN, no activation   :183-189,376
    r = F.conv2d(col, sd[p + '.head_common_layers.0.weight'], sd[p + '.head_common_layers.0.bias'], 1, 1)
    r = _bn(r, sd, p + '.head_common_layers.1')
    r = F.conv2d(r, sd[p + '.head_common_layers.2.weight'], sd[p + '.head_common_layers.2.bias'], 2, 1)
    row = _bn(r, sd, p + '.head_common_layers.3')                                   # [B,16,144,144]
    # orient: conv3x3 -> BN -> conv3x3   :232-237,380
    o = F.conv2d(row, sd[p + '.orient.0.weight'], sd[p + '.orient.0.bias'], 1, 1)
    o = _bn(o, sd, p + '.orient.1')
    orient = F.conv2d(o, sd[p + '.orient.2.weight'], sd[p + '.orient.2.bias'], 1, 1)
    rowp = F.pad(row, (half_buff, half_buff, 0, 0))                                 # :382
    colp = F.pad(col, (2 * half_buff, 2 * half_buff, 0, 0))                         # :383
    outs = {k: [] for k in ('proposal_conf', 'ext2', 'cls2', 'offset2', 'prop_bi_seg')}

    def mlp(tok, name):   # Conv1d -> BN1d -> Conv1d (no activation)  :206-228
        y = F.conv1d(tok, sd[f'{p}.{name}.0.weight'], sd[f'{p}.{name}.0.bias'])
        y = F.batch_norm(y, sd[f'{p}.{name}.1.running_mean'], sd[f'{p}.{name}.1.running_var'],
                         sd[f'{p}.{name}.1.weight'], sd[f'{p}.{name}.1.bias'], False, 0.0, BN_EPS)
        y = F.conv1d(y, sd[f'{p}.{name}.2.weight'], sd[f'{p}.{name}.2.bias'])
        return y.transpose(1, 2)                                                    # 'b c h -> b h c'

    for i in range(num_prop):                                                       # :390-421
        local = rowp[:, :, :, prop_width * i: prop_width * i + fw]                  # [B,16,144,10]
        upf = colp[:, :, :, 2 * prop_width * i: 2 * prop_width * i + 2 * fw]        # [B,16,288,20]
        seg = _up(F.conv2d(F.relu(upf), sd[p + '.bi_seg_proposal.weight'], sd[p + '.bi_seg_proposal.bias']),
                  h * 8, fw * 8)                                                    # :400 [B,1,1152,80]
        tok = F.avg_pool2d(seg, 8) * local                                          # :402 raw logits (quirk C2)
        tok = tok.permute(0, 2, 1, 3).reshape(B, h, -1).transpose(1, 2)             # :191-195 -> [B,160,144]
        outs['proposal_conf'].append(F.linear(tok.reshape(B, -1), sd[p + '.proposal_confidence.1.weight'],
                                              sd[p + '.proposal_confidence.1.bias']))
        outs['ext2'].append(mlp(tok, 'ext2'))
        outs['cls2'].append(mlp(tok, 'cls2'))
        outs['offset2'].append(mlp(tok, 'offset2'))
        if want_prop_bi_seg:
            outs['prop_bi_seg'].append(seg)
    res = {k: torch.stack(v, dim=1) for k, v in outs.items() if v}
    res['orient'] = orient
    return res


def detector_forward(sd, proj, vit_seg=True):
    """Detector1stage.forward up to the raw head outputs (detector1stage.py:25-51)."""
    fea, fea_up, bi_seg, endp = fpn_forward(sd, proj)
    if vit_seg:
        fea = vit_forward(sd, fea)
    out = head_forward(sd, fea, fea_up)
    out['semantic_seg'] = bi_seg
    out['endp_est'] = endp
    return out
