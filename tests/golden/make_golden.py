#!/usr/bin/env python3
"""Generate golden vectors by running the upstream reference (read-only, /root/reference)
in the build container.  Committed outputs: tests/golden/*.npz (+ g9_lanes.json).

    python tests/golden/make_golden.py [g2 g3 g4 g5 g6 g7 g9 g10]

Inputs and weights come from the repo-owned seeded generators (lanemapping_amd/synth.py,
tests/golden/cases.py), so fixtures store seeds + expected outputs, never weights.
`np.argsort` / `torch.argsort` are forced to their *stable* variants inside this process:
the reference's default unstable sorts make its own output host-dependent on tied keys
(SURVEY.md C16/C17); the build defines ties -> lower index.  For G6 the reference is also
run with its default sorts and agreement is recorded per case (`ref_default_agrees`).
"""
import io
import json
import os
import sys
import contextlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import _refload  # noqa: E402
import cases  # noqa: E402
from lanemapping_amd import synth  # noqa: E402

_np_argsort = np.argsort
_t_argsort = torch.argsort


def _stable_sorts(on=True):
    if on:
        np.argsort = lambda a, axis=-1, kind=None, order=None, **kw: _np_argsort(a, axis=axis, kind='stable', order=order)
        torch.argsort = lambda x, dim=-1, descending=False, stable=False: _t_argsort(x, dim=dim, descending=descending, stable=True)
    else:
        np.argsort = _np_argsort
        torch.argsort = _t_argsort


def _quiet(fn, *a, **k):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **k)


def ref_net(config='configs/Proj_polyline_fpn_vit_vertex_2.py', seed=2021, **over):
    over.setdefault('is_gt_avai', False)
    cfg = _refload.load_cfg(config, **over)
    net = _quiet(_refload.build_ref_net, cfg)
    synth.fill_module_(net, seed)
    return cfg, net


def save(name, **arrs):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrs)
    print(f'wrote {name}: {os.path.getsize(path) / 1e6:.2f} MB')


def g2(cfg, net):
    x = torch.from_numpy(synth.bev_batch([2021, 2022], 288))
    with torch.no_grad():
        fea, fea_up, bi_seg, endp = net.pcencoder({'proj': x})
    save('g2_fpn.npz', seeds=np.array([2021, 2022]), size=288, weight_seed=2021,
         fea=fea.numpy(), fea_up=fea_up.numpy(), bi_seg=bi_seg.numpy(), endp=endp.numpy())


def g3(cfg, net):
    x = torch.from_numpy(cases.vit_input(31))
    with torch.no_grad():
        y = net.backbone(x)
    save('g3_vit.npz', input_seed=31, weight_seed=2021, out=y.numpy())


def g4(cfg, net):
    x, x_up = cases.head_inputs(41)
    with torch.no_grad():
        out = net.heads(torch.from_numpy(x), torch.from_numpy(x_up), torch.zeros(1, 1, 1152, 1152))
    keep = {k: out[k].numpy() for k in ('proposal_conf', 'ext2', 'cls2', 'offset2', 'orient')}
    top2 = torch.topk(out['cls2'], 2, dim=-1).values
    keep['cls2_margin'] = (top2[..., 0] - top2[..., 1]).numpy()
    top2 = torch.topk(out['orient'], 2, dim=1).values
    keep['orient_margin'] = (top2[:, 0] - top2[:, 1]).numpy()
    save('g4_head.npz', input_seed=41, weight_seed=2021, **keep)


def _decode_ref(cfg, net, raw):
    out = {k: torch.from_numpy(np.array(v)) for k, v in raw.items()}
    out['prop_bi_seg'] = torch.zeros(1)
    net.heads.b_size = out['cls2'].shape[0]
    return net.heads.get_exist_coor_endp_dict(out)


def g5(cfg, net):
    raw = cases.decode_inputs(51, batch=2)
    scores = torch.sigmoid(torch.from_numpy(raw['endp_est'][:, 0, 20:-20, 20:-20])).reshape(2, -1)
    top = torch.sort(scores, dim=1, descending=True).values[:, :520]
    assert bool((top[:, 1:] < top[:, :-1]).all()), 'top-520 endpoint scores must be distinct'
    d = _decode_ref(cfg, net, raw)
    eh = [np.stack(np.nonzero(d['endp'][b].numpy()), axis=1) for b in range(2)]
    save('g5_decode.npz', input_seed=51, batch=2,
         prop_conf=d['prop_conf'].numpy(), prop_v_ext=d['prop_v_ext'].numpy().astype(np.uint8),
         prop_cls_conf=d['prop_cls_conf'].numpy(), cls_offset=d['cls_offset'].numpy(),
         orient=d['orient'].numpy().astype(np.uint8), semantic_seg=d['semantic_seg'].numpy().astype(np.uint8),
         bi_seg_rows=d['bi_seg'].numpy()[:, 3::8, :], bi_seg_sum=d['bi_seg'].numpy().astype(np.float64).sum(axis=(1, 2)),
         endp0=eh[0], endp1=eh[1])


def _postproc_ref(cfg, net, case):
    """Run the reference's get_lane_map_numpy_with_label on one hand-built decode result."""
    out = {
        'prop_conf': torch.from_numpy(np.stack([1 - case['prop_conf1'], case['prop_conf1']], axis=1)[None].astype(np.float32)),
        'prop_v_ext': torch.from_numpy(case['prop_v_ext'][None].astype(np.float32)),
        'prop_cls_conf': torch.zeros(1, 72, 144, 10),
        'cls_offset': torch.from_numpy(case['cls_offset'][None].astype(np.float64)),
        'orient': torch.full((1, 144, 144), 5, dtype=torch.int64),
        'bi_seg': torch.from_numpy(cases.expand_rows(case['bi_seg_rows'])[None]),
        'endp': torch.from_numpy(cases.endp_map(case['endp_pts'])[None]),
    }
    maps = net.heads.get_lane_map_numpy_with_label(out, {}, is_flip=False, is_img=False,
                                                   is_get_1_stage_result=False, is_gt_avai=False)
    V = maps['cls_offset_smooth'][0]
    E = np.stack(np.nonzero(maps['endp_by_cls'][0]), axis=1)
    return V, E


def g6(cfg, net):
    res = {}
    n = cases.NUM_POSTPROC_CASES
    agree = np.zeros(n, dtype=np.uint8)
    for i in range(n):
        case = cases.postproc_case(i)
        _stable_sorts(True)
        V, E = _postproc_ref(cfg, net, case)
        _stable_sorts(False)
        V2, E2 = _postproc_ref(cfg, net, case)
        _stable_sorts(True)
        # row order of the 72 output slots is itself sort-dependent: compare as sets of lines
        def canon(A):
            return A[np.lexsort(A.reshape(A.shape[0], -1).T[::-1])]
        agree[i] = int(np.array_equal(canon(V), canon(V2)) and np.array_equal(E, E2))
        res[f'V{i}'] = V
        res[f'E{i}'] = E.astype(np.int32)
        nl = int((np.count_nonzero(V[:, :, 0] > 0, axis=1) >= 2).sum())
        print(f'  case {i}: {nl} lines, {len(E)} endpoints, default-sort agrees={agree[i]}')
    save('g6_postproc.npz', n_cases=n, ref_default_agrees=agree, **res)
    # G9: JSON text of save_lane_seq_2d for case 0 (utils/io_utils.py:58-93)
    from baseline.utils.io_utils import save_lane_seq_2d
    V = res['V0']
    packed = np.zeros((72, 144, 3))
    packed[:, :, 0] = np.arange(3, 1152, 8)
    packed[:, :, 1:] = V
    path = os.path.join(HERE, 'g9_lanes.json')
    save_lane_seq_2d(packed, path)
    print('wrote g9_lanes.json', os.path.getsize(path))


def g7(cfg, net):
    cfg1, net1 = ref_net('configs/Proj_FPN_Seg.py', view=False)
    raw = cases.decode_inputs(71, batch=1)
    pred = {'seg': torch.from_numpy(raw['semantic_seg']), 'endp': torch.from_numpy(raw['endp_est'])}
    r = net1.pcencoder.infer_validate(pred, seg_thre=cfg1.seg_thre, endp_thre=cfg1.endp_thre)
    save('g7_segmentor.npz', input_seed=71, seg=r['seg'].numpy().astype(np.uint8),
         endp=np.stack(np.nonzero(r['endp'][0].numpy()), axis=1))


def g10(cfg, net):
    """End-to-end: one 1152² synthetic tile through the whole reference net (config 2)."""
    x = torch.from_numpy(synth.bev_batch([2021], 1152))
    cap = {}

    def hook(mod, args, out):
        cap['raw'] = {k: v.detach().clone() for k, v in out.items() if k != 'prop_bi_seg' and k != 'endpoint'}
    hd = net.heads.register_forward_hook(hook)
    orig = net.heads.get_exist_coor_endp_dict

    def spy(out):
        cap['sem_logits'] = out['semantic_seg'].detach().clone()
        d = orig(out)
        cap['dec'] = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in d.items()}
        return d
    net.heads.get_exist_coor_endp_dict = spy
    with torch.no_grad():
        o = net({'proj': x})
    hd.remove()
    raw, dec = cap['raw'], cap['dec']
    V = o['lane_maps']['cls_offset_smooth'][0]
    top2 = torch.topk(raw['cls2'], 2, dim=-1).values
    # decision margins of the thresholded outputs, so a consumer can tell an fp32-noise flip from a bug:
    # flat indices of pixels / rows whose class decision sits within 1e-4 of a tie or of the threshold
    sm = cap['sem_logits'].softmax(1)[0]
    s1, s2 = sm[1], sm[2]
    sem_margin = torch.minimum((s1 - s2).abs(), (torch.maximum(s1, s2) - cfg.coor_thre).abs())
    e = raw['ext2'].softmax(3)[0]
    ext_margin = torch.minimum((e[..., 1] - e[..., 2]).abs(), (torch.maximum(e[..., 1], e[..., 2]) - cfg.exist_thre).abs())
    otop = torch.topk(raw['orient'], 2, dim=1).values[0]
    save('g10_e2e.npz', tile_seed=2021, weight_seed=2021,
         proposal_conf=raw['proposal_conf'].numpy(), ext2=raw['ext2'].numpy(), cls2=raw['cls2'].numpy(),
         offset2=raw['offset2'].numpy(), orient_logits=raw['orient'].numpy(),
         cls2_margin=(top2[..., 0] - top2[..., 1]).numpy(),
         sem_lowmargin=torch.nonzero(sem_margin.flatten() < 1e-4).flatten().numpy().astype(np.int64),
         ext_lowmargin=torch.nonzero(ext_margin.flatten() < 1e-4).flatten().numpy().astype(np.int64),
         orient_lowmargin=torch.nonzero((otop[0] - otop[1]).flatten() < 1e-4).flatten().numpy().astype(np.int64),
         prop_conf=dec['prop_conf'].numpy(), prop_v_ext=dec['prop_v_ext'].numpy().astype(np.uint8),
         cls_offset=dec['cls_offset'].numpy(), orient=dec['orient'].numpy().astype(np.uint8),
         semantic_seg=dec['semantic_seg'].numpy().astype(np.uint8),
         bi_seg_rows=dec['bi_seg'].numpy()[:, 3::8, :],
         endp=np.stack(np.nonzero(dec['endp'][0].numpy()), axis=1),
         endp_final=np.stack(np.nonzero(o['lane_maps']['endp_by_cls'][0]), axis=1),
         cls_offset_smooth=V)


def g8(cfg, net):
    """RowRef head (config 4): forward incl. the shrinking-range scatter, decode, label-free line assembly."""
    cfg4 = _refload.load_cfg('configs/Proj28_GFC-T3_RowRef_82_73_laser.py', vit_seg=True, is_gt_avai=False)
    from baseline.models.registry import build_heads
    torch.manual_seed(2021)
    head = build_heads(cfg4).eval()
    synth.fill_module_(head, 2021, prefix='heads.')
    x = cases.head_inputs(81, batch=2)[0]
    with torch.no_grad():
        out = head(torch.from_numpy(x))
        dec = head.get_exist_coor_endp_dict(out)
    keep = {}
    for c in range(12):
        keep[f'ext_mean_{c}'] = out[f'ext_{c}'][:, :, 0].mean(dim=1).numpy()
        keep[f'ext2_{c}'] = out[f'ext2_{c}'].numpy()
        keep[f'cls2_arg_{c}'] = out[f'cls2_{c}'].argmax(dim=2).numpy().astype(np.int16)
        top2 = torch.topk(out[f'cls2_{c}'], 2, dim=2).values
        keep[f'cls2_margin_{c}'] = (top2[..., 0] - top2[..., 1]).numpy()
        keep[f'cls2_max_{c}'] = top2[..., 0].numpy()
    conf, cls = dec['conf'].numpy(), dec['cls'].numpy()
    # label-free part of get_lane_map_numpy_with_label (:487-516)
    from baseline.utils.polyline_utils import smooth_cls_line_per_batch
    lines = []
    for b in range(2):
        conf_pred = np.where(conf[b] > cfg4.conf_thr, 1, 0)
        cls_idx = np.argmax(torch.nn.functional.softmax(torch.from_numpy(cls[b]), dim=0).numpy(), axis=0)
        cls_idx[np.where(cls_idx == 12)] = 255
        cci = cls_idx.copy()
        cci[np.where(conf_pred == 0)] = 255
        pl = np.zeros((12, 144)) - 1.0
        for l in range(12):
            px = np.where(cci == l)
            pl[l, px[0]] = px[1] / 144 * 1152. + 4
        lines.append(smooth_cls_line_per_batch(pl, np.zeros((144, 144)) + 5, complete_inner_nodes=True))
    # second run with thr_ext = 0.5 so that only part of the lanes goes through the token transformer
    head.thr_ext = 0.5
    with torch.no_grad():
        out5 = head(torch.from_numpy(x))
        dec5 = head.get_exist_coor_endp_dict(out5)
    keep['t5_selected'] = np.array([[float(out5[f'ext_{c}'][b, :, 0].mean()) > 0.5 for c in range(12)] for b in range(2)])
    keep['t5_conf'] = dec5['conf'].numpy().astype(np.uint8)
    keep['t5_cls'] = dec5['cls'].numpy().astype(np.uint8)
    for c in range(12):
        keep[f't5_ext2_{c}'] = out5[f'ext2_{c}'].numpy()
    save('g8_rowref.npz', input_seed=81, weight_seed=2021, conf=conf.astype(np.uint8), cls=cls.astype(np.uint8),
         pred_lines=np.stack(lines), **keep)


def g11(cfg, net):
    """In-repo tail of LidarEncoder.forward (lidarencoder.py:70-81): flip, bicubic, fea_aligner, fea_conv, 1x1 heads,
    bilinear.  mmdet3d is absent, so the voxeliser / SparseEncoder are replaced by nn.Identity at construction and the
    backbone's dense output is injected (extract_lidar_feat patched) - those two stay PARITY UNPINNED."""
    import importlib
    import types
    _refload.install()
    le_mod = importlib.import_module('baseline.models.pcencoder.lidarencoder')
    le_mod.VoxelizationByGridShape = lambda **k: torch.nn.Identity()
    le_mod.MODELS = types.SimpleNamespace(build=lambda c: torch.nn.Identity())
    D = _refload._AttrDict
    lcfg = D(gt_downsample_ratio=8)
    lidar_encoder = D(voxelize=dict(point_cloud_range=[-15., -25., -2., 15., 25., 2.], max_num_points=10,
                                    grid_shape=[96, 96, 10], max_voxels=100000),
                      backnone=dict(type='SparseEncoder', in_channels=4, sparse_shape=[21, 100, 100], output_channels=128))
    m = le_mod.LidarEncoder(Xn=24, Yn=24, out_channels=64, lidar_encoder=lidar_encoder, cfg=lcfg).eval()
    synth.fill_module_(m, 2021, prefix='pcencoder.')
    dense = cases.lidar_tail_input(111)
    m.extract_lidar_feat = lambda pts: torch.from_numpy(dense)
    with torch.no_grad():
        fea, fea_up, bi, en = m({'points': [types.SimpleNamespace(data=None)] * dense.shape[0]})
    save('g11_lidar_tail.npz', input_seed=111, weight_seed=2021, Xn=24, Yn=24, fea=fea.numpy(), fea_up=fea_up.numpy(),
         bi_seg=bi.numpy().astype(np.float32), endp=en.numpy().astype(np.float32))


def g12(cfg, net):
    """f1: transform_coordinate_from_img_2_pc of the reference (baseline/utils/coor_img2pc.py) on 3 seeded cases."""
    _refload.install()
    sys.path.insert(0, os.path.join(_refload.REF_ROOT, 'baseline', 'utils'))
    import coor_img2pc as ref
    keep = {}
    for i, seed in enumerate((501, 502, 503)):
        params, seqs, lens, tile = cases.img2pc_case(seed)
        keep[f'out_{i}'] = ref.transform_coordinate_from_img_2_pc(params, seqs.copy(), list(lens), tile.copy())
    # file-level driver (:185-220): 2-D JSON + PNG + parameter file -> 3-D JSON / TXT
    import tempfile
    from PIL import Image
    with tempfile.TemporaryDirectory() as d:
        cases.write_img2pc_files(d, 501)
        ref.transform_coordinate_from_img_2_pc_single(f'{d}/t.json', f'{d}/t.png', f'{d}/t.txt', f'{d}/o.json', f'{d}/o.txt')
        keep['file_json'] = np.array(open(f'{d}/o.json').read())
        keep['file_txt'] = np.array(open(f'{d}/o.txt').read())
    save('g12_img2pc.npz', seeds=np.array([501, 502, 503]), **keep)


def g13(cfg, net):
    """f2: merge_lines + downsample_seqs of the reference (baseline/utils/merge_lines.py) on a seeded 5-tile road."""
    import tempfile
    _refload.install()
    sys.path.insert(0, os.path.join(_refload.REF_ROOT, 'baseline', 'utils'))
    import merge_lines as ref
    keep = {}
    with tempfile.TemporaryDirectory() as d:
        merged = _quiet(ref.merge_lines, cases.merge_case_files(d))
    keep['n'] = np.array(len(merged))
    for i, sq in enumerate(merged):
        keep[f'merged_{i}'] = sq
        keep[f'down_{i}'] = ref.downsample_seqs(sq)
    save('g13_merge.npz', **keep)
    print('merged lines:', [len(m) for m in merged])


def g14(cfg, net):
    """f3: cal_coor_measures ('conf') and eval_metric_endp_detector of the reference on seeded pred / GT pairs."""
    _refload.install()
    from baseline.utils import metric_utils as ref
    keep = {}
    for i, seed in enumerate((801, 802, 803, 804)):
        label, pred, egt, epr = cases.metric_case(seed)
        keep[f'coor_{i}'] = np.array(ref.cal_coor_measures(label, pred, 'conf', offset_thre=8 if i % 2 else 16), dtype=np.float64)
        keep[f'endp_{i}'] = np.array(ref.eval_metric_endp_detector(epr, egt, r_thre=10), dtype=np.float64)
    save('g14_metrics.npz', seeds=np.array([801, 802, 803, 804]), **keep)


G15_GAINS = {'heads.offset2.2.weight': 0.02, 'heads.offset2.2.bias': 0.02}     # |offset2| stays inside one bin
G15_SCREEN = tuple(range(2021, 2041))      # tile seeds offered to the stability screen (round 4: 20 instead of 3)
G15_KEEP = 10
G17_SCREEN = tuple(range(3101, 3113))      # LAS cloud seeds offered to the screen of the headline-chain golden
G17_KEEP = 4
G17_RASTER = dict(local_min_ele=-0.5, ele_reso=0.02)                              # bench.py's rasteriser parameters


DENSE_KEEP = 4          # tiles per stability-screened golden that also carry prop_cls_conf and bi_seg rows (round 6)


def _stable_net():
    """The reference net with the G15 gains, and a spy that keeps a copy of what the decode returned (the post-processing mutates it)."""
    cfg2, net2 = ref_net(seed=2021)
    synth.apply_gains_(net2, G15_GAINS)
    cap = {}
    orig = net2.heads.get_exist_coor_endp_dict

    def spy(out):
        cap['cls2'] = out['cls2'].detach().clone()          # raw column-bin logits [B, proposals, rows, bins] (round 6: their margin is stored)
        d = orig(out)
        cap['dec'] = {k: (v.detach().clone() if torch.is_tensor(v) else v) for k, v in d.items()}
        return d
    net2.heads.get_exist_coor_endp_dict = spy
    return net2, cap


def _screen(net2, cap, x, noise_key):
    """The reference's final output for x, or None when it is not identical for x + 1e-5 n and x - 1e-5 n (n = seeded +-1 noise):
    which vertices exist, their semantics, the kept endpoints; columns within 1e-2 px."""
    def run(t):
        with torch.no_grad():
            return _quiet(net2, {'proj': t})
    noise = torch.from_numpy(((synth.uniform(noise_key, x.numel()) > 0.5).astype(np.float32) * 2 - 1).reshape(x.shape))
    o = run(x)
    dec = cap['dec']
    # MARGIN of the column-bin decisions (round 6): cls_offset = bin + offset jumps by whole bins where the top two cls2 logits are a
    # near-tie.  The 1e-5 screen only guards cells that end up in a polyline; elsewhere (existence class 2 cells outside every line) a
    # path that is within 1e-4 of the reference's logits may pick the other bin (golden G17, cloud 3103, proposal 63 row 107: margin
    # 2.1e-5, taken by the direct-convolution route).  Stored so that the test can hold cls_offset to 1e-4 EXCEPT in cells whose
    # reference margin is below the tolerance - the rule G10 applies to its class flips.
    top2 = torch.topk(cap['cls2'][0], 2, dim=-1).values
    cls_margin = (top2[..., 0] - top2[..., 1]).numpy().astype(np.float32)
    V = o['lane_maps']['cls_offset_smooth'][0]
    E = np.stack(np.nonzero(o['lane_maps']['endp_by_cls'][0]), axis=1)
    for sgn in (1.0, -1.0):
        o2 = run(x + sgn * 1e-5 * noise)
        V2 = o2['lane_maps']['cls_offset_smooth'][0]
        E2 = np.stack(np.nonzero(o2['lane_maps']['endp_by_cls'][0]), axis=1)
        if not (np.array_equal(V[:, :, 0] > 0, V2[:, :, 0] > 0) and np.array_equal(V[:, :, 1], V2[:, :, 1]) and np.array_equal(E, E2)
                and float(np.abs(V[:, :, 0] - V2[:, :, 0]).max()) < 1e-2):
            return None
    nl = int((np.count_nonzero(V[:, :, 0] > 0, axis=1) >= 2).sum())
    D = np.stack(np.nonzero(dec['endp'][0].numpy()), axis=1).astype(np.int32)
    # MARGIN of the endpoint decisions (round 4): the endpoint pick - top-K scores, <= 20 px clustering, the sample nearest the centroid
    # - is a near-tie wherever a pixel enters or leaves the top K, and the 1e-5 screen above does not see every such case (tile 2033: the
    # reference itself moves one endpoint by a pixel under a 1e-4 perturbation; so does an exact direct-convolution path).  Like the class
    # flips of G10 (reference margin < 1e-4), an endpoint may differ only where the reference's OWN decision changes under a perturbation of
    # the tolerance's size: eight more runs at 1e-4 (four noise patterns, both signs) give, per endpoint of the unperturbed run, whether it is
    # present in all of them (`*_firm`), and the union of everything any run produced (`*_any`); `lines_firm` = the polylines survive too.
    dset, eset = {tuple(r) for r in D}, {tuple(r) for r in E}
    d_firm, e_firm, d_any, e_any, lines_firm = set(dset), set(eset), set(dset), set(eset), True
    for key in (1, 2, 3, 4):
        n4 = torch.from_numpy(((synth.uniform(synth.fnv1a64('margin%d' % key) ^ noise_key, x.numel()) > 0.5).astype(np.float32) * 2 - 1).reshape(x.shape))
        for sgn in (1.0, -1.0):
            o4 = run(x + sgn * 1e-4 * n4)
            V4 = o4['lane_maps']['cls_offset_smooth'][0]
            E4 = {tuple(r) for r in np.stack(np.nonzero(o4['lane_maps']['endp_by_cls'][0]), axis=1)}
            D4 = {tuple(r) for r in np.stack(np.nonzero(cap['dec']['endp'][0].numpy()), axis=1)}
            d_firm &= D4; e_firm &= E4; d_any |= D4; e_any |= E4
            lines_firm = lines_firm and np.array_equal(V[:, :, 0] > 0, V4[:, :, 0] > 0) and np.array_equal(V[:, :, 1], V4[:, :, 1])
    as_arr = lambda st: np.array(sorted(st), dtype=np.int32).reshape(-1, 2)
    info = (f'{nl} lines, {len(E)} endpoints ({len(e_firm)} firm under 1e-4; decode: {len(D)}, {len(d_firm)} firm), lines firm: {lines_firm}, '
            f'max |cls_offset - proposal origin| {float((dec["cls_offset"][0] - (2 * torch.arange(72)[:, None] - 4)).abs().max()):.3f}')
    # (round 6) the class confidences and the foreground probability on the rows the assembly reads (8 h + 3), for absolute-1e-4 checks
    # end to end; bi_seg on every other such row (h even) to bound the fixture's size - the caller keeps them for its first tiles only
    dense = {'prop_cls_conf': dec['prop_cls_conf'][0].numpy().astype(np.float32),
             'bi_seg_rows': dec['bi_seg'][0].numpy().astype(np.float32)[3::16].copy()}
    return {'V': V, 'E': E.astype(np.int32), 'dense': dense, 'prop_conf': dec['prop_conf'][0].numpy(), 'prop_v_ext': dec['prop_v_ext'][0].numpy().astype(np.uint8),
            'cls_offset': dec['cls_offset'][0].numpy(), 'cls_margin': cls_margin, 'endp': D, 'endp_firm': as_arr(d_firm), 'endp_any': as_arr(d_any),
            'E_firm': as_arr(e_firm), 'E_any': as_arr(e_any), 'lines_firm': np.array(lines_firm)}, info


def g15(cfg, net):
    """Stability-screened end-to-end golden.  The polyline assembly is discontinuous, and on the G10 tile (seeded weights as they
    come: |offset2| up to 30) a 1e-7 change regroups vertices, so G10 pins it stage by stage only.  Here the offset-regression
    layer is scaled by 0.02 (vertex columns then stay inside their bin) and the tiles are SCREENED: kept only if the reference's
    own final output - which vertices exist, their semantics, the kept endpoints - is identical for x, x + 1e-5 n and x - 1e-5 n
    (n = seeded +-1 noise) and the columns move by < 1e-2 px.  An implementation that matches the decode within 1e-4 must then
    reproduce the reference's polylines exactly.  Round 4: tile seeds 2021 .. 2040 go through the screen, the first G15_KEEP stable
    ones are kept (round 3 kept 2021 and 2023 of three)."""
    net2, cap = _stable_net()
    keep, kept, failed = {}, [], []
    for ts in G15_SCREEN:
        if len(kept) == G15_KEEP:
            break
        x = torch.from_numpy(synth.bev_batch([ts], 1152))
        r = _screen(net2, cap, x, synth.fnv1a64('g15noise') ^ ts)
        if r is None:
            failed.append(ts)
            print(f'  tile {ts}: not stable under a 1e-5 input perturbation - skipped')
            continue
        print(f'  tile {ts}: {r[1]}')
        dense = r[0].pop('dense')
        if len(kept) < DENSE_KEEP:
            r[0].update(dense)
        for k, v in r[0].items():
            keep[f'{k}{len(kept)}'] = v
        kept.append(ts)
    save('g15_e2e_stable.npz', tile_seeds=np.array(kept), screened_out=np.array(failed), weight_seed=2021,
         gain_keys=np.array(list(G15_GAINS)), gain_values=np.array(list(G15_GAINS.values())), **keep)


def g17(cfg, net):
    """The HEADLINE chain as one golden (round 4): a seeded 4,194,304-point LAS-shaped cloud (synth.las_points) is rasterised by the C
    oracle (oracle/raster_ref.c, bench.py's parameters), the REFERENCE net (G15 gains) runs on u8 / 255 of that tile - the load_img
    contract, laserlane_proposals.py:85-98 - and the result goes through the G15 stability screen.  The GPU test feeds the same points
    to lm_bev_raster_batch -> TilePipeline and must reproduce the reference's polylines.  (The rasteriser itself stays parity-unpinned:
    the reference has none; what this pins is the composition raster -> u8 tile -> stem -> ... -> polylines.)"""
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
    from oracle import raster_ref
    net2, cap = _stable_net()
    keep, kept, failed = {}, [], []
    for cs in G17_SCREEN:
        if len(kept) == G17_KEEP:
            break
        pts = synth.las_points(cs)
        u8 = raster_ref.raster(pts, raster_ref.params(**G17_RASTER), 1152, 1152)
        x = torch.from_numpy((u8.astype(np.float32) / np.float32(255.0)).transpose(2, 0, 1).copy())[None]
        r = _screen(net2, cap, x, synth.fnv1a64('g17noise') ^ cs)
        if r is None:
            failed.append(cs)
            print(f'  cloud {cs}: not stable under a 1e-5 input perturbation - skipped')
            continue
        print(f'  cloud {cs}: {r[1]}, {int((u8[:, :, 0] > 0).sum())} occupied pixels')
        dense = r[0].pop('dense')
        if len(kept) < DENSE_KEEP:
            r[0].update(dense)
        for k, v in r[0].items():
            keep[f'{k}{len(kept)}'] = v
        keep[f'tile_crc{len(kept)}'] = np.array(int(np.frombuffer(u8.tobytes(), dtype=np.uint8).astype(np.uint64).sum()))
        kept.append(cs)
    save('g17_chain.npz', cloud_seeds=np.array(kept), screened_out=np.array(failed), weight_seed=2021, n_points=4194304,
         raster_keys=np.array(list(G17_RASTER)), raster_values=np.array(list(G17_RASTER.values())),
         gain_keys=np.array(list(G15_GAINS)), gain_values=np.array(list(G15_GAINS.values())), **keep)


def g16(cfg, net):
    """f4: label-JSON schema - load_seq / save_seq / cal_seq_orientation of the reference (data/convert_data.py:25-70, :72-) on a seeded file."""
    import tempfile
    _refload.install()
    sys.path.insert(0, os.path.join(_refload.REF_ROOT, 'data'))
    import convert_data as ref
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, 'label.json')
        open(src, 'w').write(cases.label_json_text(1601))
        seq, lens, sem, inst, init, end = ref.load_seq(src)
        orient = ref.cal_seq_orientation(seq, lens)
        out = os.path.join(d, 'out.json')
        _quiet(ref.save_seq, seq, lens, sem, inst, orient, out)
        text = open(out).read()
    save('g16_label_json.npz', seed=1601, seq=seq, seq_lens=np.array(lens), semantic=np.array(sem), instance=np.array(inst),
         init=np.array(init, dtype=np.float64), end=np.array(end, dtype=np.float64), orient=np.asarray(orient), saved_text=np.array(text))


def g18(cfg, net):
    """a12 / f3: the dataset contract.  A synthetic <data_root> (cases.write_dataset: split file, cropped_tiff/, labels/sparse_*/) is
    listed by the reference's LaserLaneProposal for every mode (the 'test' list right after random.seed(2021), i.e. in the order the
    reference's Runner gets it: runner.py:69-71, laserlane_proposals.py:512-514), and format_gt_column_proposal gives the evaluation
    ground truth of every test tile.  skimage is absent: block_reduce (feeds only the training label 'instance', not stored here)
    is replaced by a numpy block maximum for this golden."""
    import random
    import tempfile
    _refload.install()
    import baseline.datasets.laserlane_proposals as lp

    def block_reduce(a, block_size=8, func=np.max):
        h, w = a.shape
        return func(a.reshape(h // block_size, block_size, w // block_size, block_size), axis=(1, 3))
    lp.skimage.measure.block_reduce = block_reduce
    rcfg = _refload.load_cfg('configs/Proj_polyline_fpn_vit_vertex_2.py')
    keep = {}
    with tempfile.TemporaryDirectory() as d:
        cases.write_dataset(d, n_tiles=G18_TILES, seed=G18_SEED)
        for mode in ('test', 'valid', 'single', 'all', 'infer_only', 'train'):
            random.seed(2021)
            ds = lp.LaserLaneProposal(d, 'data_split-shuffle.json', mode=mode, cfg=rcfg)
            keep['stems_' + mode] = np.array(ds.image_stem_list)
            keep['images_' + mode] = np.array([os.path.relpath(p, d) for p in ds.image_list])
            if mode == 'test':
                keep['seq_test'] = np.array([os.path.relpath(p, d) for p in ds.seq_list])
                keep['mask_test'] = np.array([os.path.relpath(p, d) for p in ds.mask_list])
                keep['instance_test'] = np.array([os.path.relpath(p, d) for p in ds.instance_list])
                keep['endp_test'] = np.array([os.path.relpath(p, d) for p in ds.endp_list])
                for i in range(len(ds)):
                    sp = ds.format_gt_column_proposal(i)
                    keep[f'lc_coor_raw_{i}'] = sp['lc_coor_raw'].numpy()
                    m = np.asarray(sp['mask'])
                    keep[f'mask_nz_{i}'] = np.concatenate([np.argwhere(m != 0), m[m != 0][:, None]], axis=1).astype(np.int32)
                    keep[f'endp_nz_{i}'] = np.argwhere(sp['endp_map'].numpy() > 0).astype(np.int32)
                    keep[f'endp_val_{i}'] = sp['endp_map'].numpy()[sp['endp_map'].numpy() > 0]
                print('  test order:', list(ds.image_stem_list))
                print('  GT vertices per lane, tile 0:', (keep['lc_coor_raw_0'] > 0).sum(1))
    save('g18_dataset.npz', n_tiles=G18_TILES, seed=G18_SEED, shuffle_seed=2021, **keep)


G18_TILES, G18_SEED = 7, 1801


def g19(cfg, net):
    """f3: the accumulation of the test loop (runner.py:741-787, :843-867) over seeded prediction / ground-truth pairs: per tile the
    reference's cal_coor_measures ('conf') and eval_metric_endp_detector (r_thre = 2 x validate_buffer), the counters summed over
    the tiles and turned into precision / recall / F1 with the loop's own formulas (EPS = 1e-16)."""
    _refload.install()
    from baseline.utils import metric_utils as ref
    EPS = 1e-16
    buf = 10
    tot = np.zeros(8)
    for seed in G19_SEEDS:
        label, pred, egt, epr = cases.metric_case(seed)
        _, _, _, TPs, seg_pts, DGs, gt_pts = ref.cal_coor_measures(label, pred, 'conf', offset_thre=buf)
        _, _, _, TP2, dets, DG2, gts = ref.eval_metric_endp_detector(epr, egt, r_thre=buf * 2)
        tot += np.array([TPs, seg_pts, DGs, gt_pts, TP2, dets, DG2, gts], dtype=np.float64)
    out = []
    for tp, seg, dg, gt in (tot[0:4], tot[4:8]):
        pre, rec, f1 = tp / (seg + EPS), dg / (gt + EPS), 0.
        if (pre + rec) > 0.:
            f1 = 2. * pre * rec / (pre + rec)
        out += [pre, rec, f1]
    save('g19_eval_loop.npz', seeds=np.array(G19_SEEDS), validate_buffer=buf, counters=tot, prf=np.array(out))
    print('  counters', tot, 'P/R/F1', out)


G19_SEEDS = (811, 812, 813, 814, 815, 816)


def main():
    which = sys.argv[1:] or ['g2', 'g3', 'g4', 'g5', 'g6', 'g7', 'g10']
    _stable_sorts(True)
    cfg, net = ref_net()
    for w in which:
        print('==', w)
        globals()[w](cfg, net)


if __name__ == '__main__':
    main()
