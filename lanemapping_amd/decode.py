"""Decode orchestration: device kernels (softmax / argmax / thresholds / top-K) + host endpoint clustering.

Mirrors ColumnProposal2.get_exist_coor_endp_dict (reference heads/polyline_fpn_vit_vertex_2.py:602-759) and
PostProjector2.infer_validate (pcencoder/postprojector.py:115-183).  `decode_compact` keeps results in the
form the polyline assembly consumes (no dense 1152x1152 endpoint map, bi_seg rows 3::8 gathered on device);
`compact_to_reference_dict` expands it to the reference's dictionary for drop-in callers.
"""
import numpy as np
import torch

from . import ops, hostpost
from ._lib import LanemapHipError

CLIP = 20
TOPK = 512       # >= largest K the reference's growth loop can reach (500 + 10)


def _cluster_all(idx, status, crop_w, k0, k_max):
    idx_h = idx.cpu().numpy()
    if int(status.cpu().max()) != 0:
        raise LanemapHipError('endpoint top-K: more tied scores than the candidate buffer holds '
                              '(saturated endpoint logits); see DESIGN.md §decode')
    out = []
    for b in range(idx_h.shape[0]):
        pts, _ = hostpost.cluster_endpoints(idx_h[b], crop_w=crop_w, clip=CLIP, k0=k0, k_max=k_max)
        out.append(pts)
    return out


def decode_compact(out, cfg, num_cls, prop_width, half_buff):
    prop_conf, v_ext, cls_conf, cls_idx, cls_offset = ops.decode_proposals(
        out['proposal_conf'], out['ext2'], out['cls2'], out['offset2'], cfg.exist_thre, prop_width, half_buff)
    orient = ops.decode_orient(out['orient'])
    sem, biseg, rows = ops.decode_semantic(out['semantic_seg'], cfg.coor_thre)
    idx, score, status = ops.endp_topk(out['endp_est'], K=TOPK, clip=CLIP)
    W = out['endp_est'].shape[-1]
    endp = _cluster_all(idx, status, W - 2 * CLIP, num_cls * 2 * 10, 500)
    return {'prop_conf': prop_conf, 'prop_v_ext': v_ext, 'prop_cls_conf': cls_conf, 'cls_idx': cls_idx,
            'cls_offset': cls_offset, 'orient': orient, 'semantic_seg_u8': sem, 'bi_seg': biseg, 'bi_seg_rows': rows,
            'endp_pts': endp, 'endp_topk_idx': idx, 'endp_topk_score': score, 'img_hw': tuple(out['endp_est'].shape[-2:])}


def endp_dense(pts_list, hw):
    m = torch.zeros((len(pts_list), hw[0], hw[1]))
    for b, pts in enumerate(pts_list):
        if len(pts):
            m[b, torch.from_numpy(pts[:, 0].astype(np.int64)), torch.from_numpy(pts[:, 1].astype(np.int64))] = 1
    return m


def compact_to_reference_dict(c):
    """Same keys / dtypes / devices as the reference's non-view_detail return (:753-757)."""
    return {'prop_conf': c['prop_conf'].cpu(), 'prop_v_ext': c['prop_v_ext'].cpu(), 'prop_cls_conf': c['prop_cls_conf'],
            'endp': endp_dense(c['endp_pts'], c['img_hw']), 'orient': c['orient'].cpu().to(torch.int64),
            'bi_seg': c['bi_seg'], 'semantic_seg': c['semantic_seg_u8'].cpu().to(torch.float32),
            'cls_offset': c['cls_offset'].cpu()}


def segmentor_decode(seg_logits, endp_logits, seg_thre):
    if seg_thre is None:
        raise NotImplementedError('seg_thre=None (plain argmax) is not used by any shipped config')
    sem, _, _ = ops.decode_semantic(seg_logits, seg_thre, raw_mode=True, want_biseg=False)
    idx, score, status = ops.endp_topk(endp_logits, K=128, clip=CLIP)
    pts = _cluster_all(idx, status, endp_logits.shape[-1] - 2 * CLIP, 6, 100)
    return {'seg': sem.cpu().to(torch.float32), 'endp': endp_dense(pts, tuple(endp_logits.shape[-2:])), 'endp_pts': pts}
