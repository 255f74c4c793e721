"""Oracle (test infrastructure): K-Lane "RowRef" head of config 4.

  rowref_forward    <- baseline/models/heads/row_shared_not_reduc_ref.py:170-246 (incl. the shrinking-range
                       scatter-back bug of :227-230, quirk C8)
  rowref_decode     <- :334-363
  rowref_pred_lines <- the label-free part of get_lane_map_numpy_with_label :487-516
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import postproc_ref


def _stack(sd, p, t):
    """Conv1d(1152->512) -> BN1d -> Conv1d(512->n), then 'b c h -> b h c' (:117-129)."""
    y = F.conv1d(t, sd[p + '.0.weight'], sd[p + '.0.bias'])
    y = F.batch_norm(y, sd[p + '.1.running_mean'], sd[p + '.1.running_var'], sd[p + '.1.weight'], sd[p + '.1.bias'], False, 0., 1e-5)
    return F.conv1d(y, sd[p + '.2.weight'], sd[p + '.2.bias']).transpose(1, 2)


def _transformer(sd, p, t, heads, dim_head):
    dim = t.shape[-1]
    B = t.shape[0]
    l = 0
    while f'{p}.layers.{l}.0.norm.weight' in sd:
        a, f = f'{p}.layers.{l}.0', f'{p}.layers.{l}.1'
        y = F.layer_norm(t, (dim,), sd[a + '.norm.weight'], sd[a + '.norm.bias'], 1e-5)
        q, k, v = [z.reshape(B, -1, heads, dim_head).transpose(1, 2) for z in F.linear(y, sd[a + '.fn.to_qkv.weight']).chunk(3, dim=-1)]
        o = torch.matmul((torch.matmul(q, k.transpose(-1, -2)) * dim_head ** -0.5).softmax(-1), v)
        o = o.transpose(1, 2).reshape(B, -1, heads * dim_head)
        t = F.linear(o, sd[a + '.fn.to_out.0.weight'], sd[a + '.fn.to_out.0.bias']) + t
        y = F.layer_norm(t, (dim,), sd[f + '.norm.weight'], sd[f + '.norm.bias'], 1e-5)
        y = F.gelu(F.linear(y, sd[f + '.fn.net.0.weight'], sd[f + '.fn.net.0.bias']))
        t = F.linear(y, sd[f + '.fn.net.3.weight'], sd[f + '.fn.net.3.bias']) + t
        l += 1
    return t


def rowref_forward(sd, x, p='heads', num_cls=12, thr_ext=0.3, off_grid=2, heads=16, dim_head=64):
    B, C, H, W = x.shape
    out = {}
    rt = x.permute(0, 1, 3, 2).reshape(B, C * W, H)                       # 'b c h w -> b (c w) h'
    for c in range(num_cls):
        out[f'ext_{c}'] = _stack(sd, f'{p}.ext_{c}', rt).softmax(2)
        out[f'cls_{c}'] = _stack(sd, f'{p}.cls_{c}', rt).softmax(2)
    pad = F.pad(x, (off_grid, off_grid)).clone()
    k = 2 * off_grid + 1
    idx_h = None                                                          # the reference's leaked loop variable
    for b in range(B):
        lanes, corr = [], []
        for c in range(num_cls):
            if out[f'ext_{c}'][b, :, 0].mean() > thr_ext:
                ci = out[f'cls_{c}'][b].argmax(dim=1)
                tok = torch.stack([pad[b, :, h, int(ci[h]):int(ci[h]) + k] for h in range(H)], dim=1)      # [C,H,k]
                idx_h = H - 1
                lanes.append(F.linear(tok.reshape(-1), sd[p + '.to_token.1.weight'], sd[p + '.to_token.1.bias']) + sd[f'{p}.emb_{c}'])
                corr.append(ci)
        if lanes:
            t = _transformer(sd, p + '.tr_lane_correlator.0', torch.stack(lanes)[None], heads, dim_head)
            t = F.layer_norm(t, (t.shape[-1],), sd[p + '.tr_lane_correlator.1.weight'], sd[p + '.tr_lane_correlator.1.bias'], 1e-5)
            t = F.linear(t, sd[p + '.tr_lane_correlator.2.weight'], sd[p + '.tr_lane_correlator.2.bias']).reshape(1, len(lanes), C, H, k)
            for i, ci in enumerate(corr):
                for h in range(idx_h):                                    # quirk C8: range shrinks by one per lane
                    pad[b, :, h, int(ci[h]):int(ci[h]) + k] = t[0, i, :, h, :]
                    idx_h = h
    x2 = pad[:, :, :, off_grid:W + off_grid]
    rt = x2.permute(0, 1, 3, 2).reshape(B, C * W, H)
    for c in range(num_cls):
        out[f'ext2_{c}'] = _stack(sd, f'{p}.ext2_{c}', rt).softmax(2)
        out[f'cls2_{c}'] = _stack(sd, f'{p}.cls2_{c}', rt).softmax(2)
    out['_refined'] = x2
    return out


def rowref_decode(out, num_cls=12):
    B, H, W = out['cls2_0'].shape
    conf = np.zeros((B, H, W))
    cls = np.zeros((B, num_cls + 1, H, W))
    for b in range(B):
        for c in range(num_cls):
            ex = out[f'ext2_{c}'][b].argmax(dim=1).numpy()
            col = out[f'cls2_{c}'][b].argmax(dim=1).numpy()
            rows = np.nonzero(ex == 0)[0]
            cls[b, c, rows, col[rows]] = 1.
            cls[b, num_cls, rows, col[rows]] = 1.
    conf[cls[:, num_cls] == 1.] = 1.
    return conf, cls


def rowref_pred_lines(conf_b, cls_b, conf_thr=0.5, num_cls=12, row_size=144):
    conf_pred = np.where(conf_b > conf_thr, 1, 0)
    cls_idx = np.argmax(torch.softmax(torch.as_tensor(cls_b), dim=0).numpy(), axis=0)
    cls_idx[cls_idx == num_cls] = 255
    cls_idx[conf_pred == 0] = 255
    lines = np.zeros((num_cls, row_size)) - 1.0
    for c in range(num_cls):
        r, w = np.nonzero(cls_idx == c)
        lines[c, r] = w / row_size * 1152. + 4
    return postproc_ref.trace_lines(lines, None)
