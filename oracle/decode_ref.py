"""Oracle (test infrastructure): decode of raw head/FPN outputs.

  decode_column_proposals <- baseline/models/heads/polyline_fpn_vit_vertex_2.py:602-759
  select_endpoints        <- same file :641-688 and cluster_select_topK_pts :903-924
  segmentor_decode        <- baseline/models/pcencoder/postprojector.py:115-183

Tie rules the reference leaves to library internals are fixed here:
descending score order with ties -> lower flat index; nearest-to-centroid ties -> lower index.
"""
import numpy as np
import torch


def cluster_centres(pts_h, pts_w, radius=20):
    """DBSCAN(eps=radius, min_samples=1) == connected components of the <=radius graph,
    then per cluster the member nearest the centroid (:903-924).  Returns (n_clusters, [(h,w)...])."""
    n = len(pts_h)
    P = np.stack([np.asarray(pts_h, dtype=np.int64), np.asarray(pts_w, dtype=np.int64)], axis=1)
    parent = list(range(n))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    r2 = radius * radius
    for i in range(n):
        d = P[:i] - P[i]
        near = np.nonzero((d * d).sum(1) <= r2)[0]
        for j in near:
            ra, rb = find(int(j)), find(i)
            if ra != rb:
                parent[max(ra, rb)] = min(ra, rb)
    roots = [find(i) for i in range(n)]
    centres = []
    for root in sorted(set(roots)):                      # first-seen order == ascending root index
        idx = [i for i in range(n) if roots[i] == root]
        Q = P[idx].astype(np.float64)
        c = Q.mean(axis=0)
        d2 = ((Q - c) ** 2).sum(1)
        centres.append(tuple(int(v) for v in P[idx[int(np.argmin(d2))]]))   # argmin: first minimum
    return len(centres), centres


def select_endpoints(logit_map, k0, k_step=10, k_max=500, clip=20, radius=20, min_clusters=4):
    """logit_map [H,W] -> (endp map [H,W] f32 in {0,1}, K used).  (:647-688)

    K grows from k0 by k_step while clusters <= min_clusters and K <= k_max."""
    H, W = logit_map.shape
    crop = torch.sigmoid(torch.as_tensor(logit_map[clip:H - clip, clip:W - clip]).float()).numpy()
    flat = crop.reshape(-1)
    order = np.lexsort((np.arange(flat.size), -flat.astype(np.float64)))   # score desc, index asc
    wc = W - 2 * clip
    K = k0
    while True:
        top = order[:K]
        ncl, centres = cluster_centres(top // wc, top % wc, radius)
        if ncl > min_clusters or K > k_max:
            break
        K += k_step
    out = np.zeros((H, W), dtype=np.float32)
    for (h, w) in centres:
        out[h + clip, w + clip] = 1.0
    return out, K, order[:K]


def decode_column_proposals(out, coor_thre=0.2, exist_thre=0.2, num_cls=12, prop_width=2, half_buff=4):
    """Raw dict (proposal_conf, ext2, cls2, offset2, orient, semantic_seg, endp_est) ->
    dict(prop_conf, prop_v_ext, prop_cls_conf, endp, orient, bi_seg, semantic_seg, cls_offset)."""
    t = {k: torch.as_tensor(v) for k, v in out.items()}
    B, P, R, Wc = t['cls2'].shape
    res = {}
    res['prop_conf'] = t['proposal_conf'].softmax(2)                                  # :610
    res['orient'] = t['orient'].argmax(1)                                             # :615
    s = t['semantic_seg'].softmax(1)                                                  # :627
    sem = torch.zeros(s.shape[0], s.shape[2], s.shape[3])
    sem[(s[:, 1] > s[:, 2]) & (s[:, 1] > coor_thre)] = 1                              # :629
    sem[(s[:, 2] > s[:, 1]) & (s[:, 2] > coor_thre)] = 2                              # :630
    res['semantic_seg'] = sem
    res['bi_seg'] = s[:, 1] + s[:, 2]                                                 # :631
    endp = np.zeros((B, R * 8, R * 8), dtype=np.float32)
    for b in range(B):
        endp[b], _, _ = select_endpoints(t['endp_est'][b, 0].numpy(), k0=num_cls * 2 * 10)
    res['endp'] = torch.from_numpy(endp)
    e = t['ext2'].softmax(3)                                                          # :694
    v = torch.zeros(B, P, R)
    v[(e[..., 1] > e[..., 2]) & (e[..., 1] > exist_thre)] = 1                         # :696
    v[(e[..., 2] > e[..., 1]) & (e[..., 2] > exist_thre)] = 2                         # :697
    res['prop_v_ext'] = v
    c = t['cls2'].softmax(-1)                                                         # :701
    res['prop_cls_conf'] = c
    idx = c.argmax(-1)                                                                # :702
    off = torch.gather(t['offset2'], 3, idx.unsqueeze(-1)).squeeze(-1)
    co = (idx.to(torch.float32) + off).to(torch.float64)                              # :726 fp32 sum kept in f64
    co = torch.where(co > Wc, torch.full_like(co, float(Wc)), co)                     # :732
    co = co + (prop_width * torch.arange(P, dtype=torch.float64) - half_buff).view(1, P, 1)   # :738
    res['cls_offset'] = co
    res['cls_idx'] = idx
    return res


def segmentor_decode(seg_logits, endp_logits, seg_thre=0.1, k0=6, k_max=100):
    """PostProjector2.infer_validate (postprojector.py:115-183): thresholds RAW logits (quirk C11)."""
    seg = torch.as_tensor(seg_logits)
    B = seg.shape[0]
    sem = torch.zeros(B, seg.shape[2], seg.shape[3])
    sem[(seg[:, 1] > seg[:, 2]) & (seg[:, 1] > seg_thre)] = 1
    sem[(seg[:, 2] > seg[:, 1]) & (seg[:, 2] > seg_thre)] = 2
    e = torch.as_tensor(endp_logits)
    endp = np.zeros((B, e.shape[2], e.shape[3]), dtype=np.float32)
    for b in range(B):
        endp[b], _, _ = select_endpoints(e[b, 0].numpy(), k0=k0, k_max=k_max)
    return {'seg': sem, 'endp': torch.from_numpy(endp)}
