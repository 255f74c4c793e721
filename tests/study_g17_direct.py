"""Study (GPU box): where does a 3x3 route differ from the reference on golden G17, and is the difference a near-tie of the REFERENCE?
Run under the route to study, e.g. `LANEMAP_WINO_F44=0 python tests/study_g17_direct.py`.  For every G17 cloud: product decode outputs vs
the golden (cls_offset / prop_conf errors, where they exceed 1e-4), and at every (proposal, row) whose column bin differs from the oracle
chain's: the oracle's own cls2 top-1 / top-2 margin there and the product's logit errors.  (Checker-side script: imports oracle/.)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, 'golden'))

from lanemapping_amd import ops, synth                                    # noqa: E402
from lanemapping_amd.boundary import build_net_from_config               # noqa: E402
from oracle import net_ref                                               # noqa: E402

if __name__ == '__main__':
    dev = torch.device('cuda:0')
    g = np.load(os.path.join(HERE, 'golden', 'g17_chain.npz'))
    net = build_net_from_config('Proj_polyline_fpn_vit_vertex_2', device='cpu')
    synth.fill_module_(net, 2021)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    for k, gain in zip(g['gain_keys'], g['gain_values']):
        sd[str(k)] = sd[str(k)] * float(gain)
    net.load_state_dict(sd, strict=True)
    net = net.to(dev)
    seeds = [int(s) for s in g['cloud_seeds']]
    clouds = [synth.las_points(s, int(g['n_points'])) for s in seeds]
    offs = np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).tolist()
    kw = {str(k): float(v) for k, v in zip(g['raster_keys'], g['raster_values'])}
    tiles = torch.empty((len(seeds), 1152, 1152, 3), device=dev, dtype=torch.uint8)
    ops.bev_raster_batch(torch.from_numpy(np.concatenate(clouds)).to(dev), offs, [ops.make_raster_params(**kw)] * len(seeds), out_u8=tiles, u8_only=True)
    with torch.no_grad():
        raw = {k: v.float().cpu() for k, v in net.forward_raw({'proj': tiles}).items()}
        o = net({'proj': tiles})
    c = {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in net.heads._compact.items()}
    print('route:', 'direct (LANEMAP_WINO_F44=0)' if os.environ.get('LANEMAP_WINO_F44', '1') == '0' else 'F(4x4) default')
    for i, s in enumerate(seeds):
        x = (tiles[i].cpu().float() / 255.0).permute(2, 0, 1)[None].contiguous()
        with torch.no_grad():
            ref = net_ref.detector_forward(sd, x)
        off_err = np.abs(c['cls_offset'][i].numpy() - g[f'cls_offset{i}'])
        conf_err = float(np.abs(c['prop_conf'][i].numpy() - g[f'prop_conf{i}']).max())
        print(f'cloud {s}: cls_offset max err {off_err.max():.3e} ({int((off_err > 1e-4).sum())} cells > 1e-4), prop_conf err {conf_err:.3e}')
        for k in ('proposal_conf', 'ext2', 'cls2', 'offset2', 'orient', 'semantic_seg', 'endp_est'):
            print(f'    raw {k:14s} max |product - oracle| {float((raw[k][i:i + 1] - ref[k]).abs().max()):.3e}   (oracle scale {float(ref[k].abs().max()):.2f})')
        mine = raw['cls2'][i].argmax(-1)
        want = ref['cls2'][0].argmax(-1)
        top = torch.topk(ref['cls2'][0], 2, dim=-1)
        for p, r in zip(*np.nonzero((mine != want).numpy())):
            a, b = int(top.indices[p, r, 0]), int(top.indices[p, r, 1])
            print(f'    column-bin flip at proposal {p} row {r}: oracle bins {a} / {b}, oracle margin {float(top.values[p, r, 0] - top.values[p, r, 1]):.3e}; '
                  f'product picks {int(mine[p, r])}; product logit errors {float(raw["cls2"][i, p, r, a] - ref["cls2"][0, p, r, a]):+.3e} / '
                  f'{float(raw["cls2"][i, p, r, b] - ref["cls2"][0, p, r, b]):+.3e}; existence class there {int(g[f"prop_v_ext{i}"][p, r])}, '
                  f'part of a golden polyline: {bool(g[f"V{i}"][p, r, 0] > 0)}')
        V = o['lane_maps']['cls_offset_smooth'][i]
        W = g[f'V{i}']
        print(f'    final polylines: vertex set equal {np.array_equal(V[:, :, 0] > 0, W[:, :, 0] > 0)}, semantics equal {np.array_equal(V[:, :, 1], W[:, :, 1])}, '
              f'max column difference {float(np.abs(V[:, :, 0] - W[:, :, 0]).max()):.3e} px')
