"""GPU parity tests (-m gpu), file 3 of 4: BASELINE.json's configs 1 / 4 / 5 end to end, the Runner (the reference's entry contract,
test_gpu_0.py) and the LAS -> map chain."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cases
from gpu_common import ROOT, _close, _lidar_module, _rowref_head
from lanemapping_amd import synth

pytestmark = pytest.mark.gpu


def test_runner_png_tiles_to_json(dev, net, tmp_path):
    """test_gpu_0.py-style entry: PNG tiles on disk -> per-tile JSON, identical to driving the pipeline directly."""
    import json
    from PIL import Image
    from lanemapping_amd import io_utils
    from lanemapping_amd.pipeline import TilePipeline
    from lanemapping_amd.runner import Runner
    seeds = [301, 302, 303]
    for s in seeds:
        Image.fromarray(synth.bev_tile_u8(s, 1152)).save(tmp_path / f'1901{s}_0001_extra.png')
    r = Runner(net.cfg, device=dev)
    r.net = net
    out = tmp_path / 'out'
    res = r.infer_lane_coordinate_endpoint_semantics(tiles=str(tmp_path), batch_size=2, work_dirs=str(out), write_lane_vertex=True)
    assert sorted(res) == [f'1901{s}_000' for s in seeds]          # image_name[0:11]
    direct = TilePipeline(net).run_batch(torch.from_numpy(synth.bev_batch(seeds, 1152)).to(dev))
    for s, (lanes, endp) in zip(seeds, direct):
        assert np.array_equal(res[f'1901{s}_000'][0], lanes)
        recs = json.load(open(out / f'1901{s}_000.json'))
        assert recs == io_utils.lane_records(io_utils.pack_lane_vertices(lanes))


def test_runner_two_ranks_byte_identical(dev, net, tmp_path):
    """Runner with torch.distributed initialised (2 ranks, gloo, both on this box's GPU): tiles are block-sharded, results are
    combined by one all-gather of f64 blocks and rank 0 writes every file - byte-identical to the single-rank run."""
    import socket
    import subprocess
    import sys
    from PIL import Image
    from lanemapping_amd.runner import Runner
    seeds = [311, 312, 313, 314, 315]                  # ragged: 3 + 2 tiles (+ 1 padding slot)
    tiles = tmp_path / 'tiles'
    tiles.mkdir()
    for s_ in seeds:
        Image.fromarray(synth.bev_tile_u8(s_, 1152)).save(tiles / f'1902{s_}_0001.png')
    r = Runner(net.cfg, device=dev)
    r.net = net
    r.infer_lane_coordinate_endpoint_semantics(tiles=str(tiles), batch_size=2, work_dirs=str(tmp_path / 'one'), write_lane_vertex=True)
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE='2', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                   LANEMAP_TEST_DEVICE='0')
        procs.append(subprocess.Popen([sys.executable, os.path.join(root, 'tests', '_runner_rank.py'), str(tiles), str(tmp_path / 'two')],
                                      env=env, cwd=root, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=900)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), '\n'.join(o[-2000:] for o in outs)
    names = sorted(os.listdir(tmp_path / 'one'))
    assert len(names) == len(seeds) and sorted(os.listdir(tmp_path / 'two')) == names
    for n in names:
        assert open(tmp_path / 'one' / n, 'rb').read() == open(tmp_path / 'two' / n, 'rb').read(), n


def test_segmentor_config1_end_to_end(dev, synth_sd):
    """BASELINE config 1 (Proj_FPN_Seg, batch 1): Segmentor through the boundary vs the oracle chain."""
    from lanemapping_amd.boundary import build_net_from_config
    from oracle import net_ref, decode_ref
    seg_net = build_net_from_config('Proj_FPN_Seg', device='cpu')
    sd = {k: v for k, v in synth_sd.items() if k.startswith('pcencoder.')}
    seg_net.load_state_dict(sd, strict=True)
    seg_net = seg_net.to(dev)
    x = torch.from_numpy(synth.bev_batch([2021], 1152))
    out = seg_net({'proj': x.to(dev)})
    with torch.no_grad():
        _, _, bi_seg, endp = net_ref.fpn_forward(synth_sd, x)
    ref = decode_ref.segmentor_decode(bi_seg.numpy(), endp.numpy(), seg_thre=0.1)
    bad = np.flatnonzero(out['seg'].numpy().reshape(-1) != ref['seg'].numpy().reshape(-1))
    l1, l2 = bi_seg[0, 1].reshape(-1)[bad], bi_seg[0, 2].reshape(-1)[bad]
    margin = torch.minimum((l1 - l2).abs(), (torch.maximum(l1, l2) - 0.1).abs())
    assert bad.size <= 32 and (bad.size == 0 or float(margin.max()) < 1e-4), f'{bad.size} seg flips, max margin {float(margin.max()) if bad.size else 0}'
    assert np.array_equal(np.stack(np.nonzero(out['endp'][0].numpy()), 1), np.stack(np.nonzero(ref['endp'][0].numpy()), 1))


def test_rowref_detector_config4_vs_oracle(dev, synth_sd):
    """Detector1stage with the RowRef head (BASELINE config 4) on one 1152^2 tile vs the oracle chain: every (lane, row) decision -
    row present (argmax ext2 == 0) and its column (argmax cls2) - equals the oracle's unless the ORACLE's own margin there is below
    1e-4; the polylines always equal the oracle's line assembly run on the product's own decode outputs."""
    from lanemapping_amd.boundary import build_net_from_config
    from oracle import net_ref, rowref_ref
    net4 = build_net_from_config('Proj28_GFC-T3_RowRef_82_73_laser', device='cpu')
    synth.fill_module_(net4, 2021)
    sd = {k: v.clone() for k, v in net4.state_dict().items()}
    for c in range(12):
        sd[f'heads.emb_{c}'] = getattr(net4.heads, f'emb_{c}').clone()
    net4 = net4.to(dev)
    x = torch.from_numpy(synth.bev_batch([2021], 1152))
    with torch.no_grad():
        o = net4({'proj': x.to(dev)})
        fea = net_ref.vit_forward(sd, net_ref.fpn_forward(sd, x)[0])
        ref = rowref_ref.rowref_forward(sd, fea)
    col_p = net4.heads._col_idx.cpu().numpy()[0]                      # [12,144]: column or -1
    flips = 0
    for c in range(12):
        e, p = ref[f'ext2_{c}'][0], ref[f'cls2_{c}'][0]               # [144,2], [144,144] probabilities
        want = np.where(e.argmax(dim=1).numpy() == 0, p.argmax(dim=1).numpy(), -1)
        top2 = torch.topk(p, 2, dim=1).values
        margin = torch.minimum((e[:, 0] - e[:, 1]).abs(), top2[:, 0] - top2[:, 1]).numpy()
        bad = np.flatnonzero(col_p[c] != want)
        flips += bad.size
        assert np.all(margin[bad] < 1e-4), f'lane {c}: decision differs from the oracle where its margin is {margin[bad].max():.2e}'
    print(f'config 4: {flips} of {12 * 144} (lane, row) decisions flipped inside the oracle margin')
    assert flips <= 8
    conf_p, cls_p = o['conf'].numpy(), o['cls'].numpy()
    assert np.array_equal(o['lane_maps']['cls_offset_smooth'][0], rowref_ref.rowref_pred_lines(conf_p[0], cls_p[0]))
    if flips == 0:
        conf, cls = rowref_ref.rowref_decode(ref)
        assert np.array_equal(conf_p, conf) and np.array_equal(cls_p, cls)


def test_lidar_encoder_forward_vs_oracle(dev):
    import cases
    from oracle import lidar_ref
    cfg = cases.small_lidar_cfg()
    m, sd = _lidar_module(dev, cfg)
    pts = [synth.lidar_points(51, 50000), synth.lidar_points(52, 30000)]
    pc = dict(cfg.pcencoder)
    pc['gt_downsample_ratio'] = 8
    ref = lidar_ref.lidar_encoder_ref(pts, sd, pc)
    with torch.no_grad():
        got = m({'points': [torch.from_numpy(p).to(dev) for p in pts]})
    for name, a, b in zip(('fea', 'fea_up', 'bi_seg', 'endp'), got, ref):
        _close(a, b, 1e-4, name)


def test_detector_config5_end_to_end(dev):
    """Detector1stage on the sparse-conv path at the real config-5 sizes (grid 576x576x10, sparse shape 21x600x600):
    raw head outputs vs the oracle chain, and the full forward (decode + polylines) runs."""
    from lanemapping_amd.boundary import build_net_from_config
    from oracle import lidar_ref, net_ref
    net5 = build_net_from_config('Proj_polyline_lidarconv_vit_vertex_2', device='cpu')
    synth.fill_module_(net5, 2021)
    sd = {k: v.clone() for k, v in net5.state_dict().items()}
    net5 = net5.to(dev)
    pts = [synth.lidar_points(61, 1 << 20)]
    pc = dict(net5.cfg.pcencoder)
    pc['gt_downsample_ratio'] = 8
    sd_pc = {k[len('pcencoder.'):]: v for k, v in sd.items() if k.startswith('pcencoder.')}
    with torch.no_grad():
        fea, fea_up, bi, en = lidar_ref.lidar_encoder_ref(pts, sd_pc, pc)
        ref = net_ref.head_forward(sd, net_ref.vit_forward(sd, fea), fea_up)
        batch = {'points': [torch.from_numpy(p).to(dev) for p in pts]}
        raw = net5.forward_raw(batch)
        _close(raw['semantic_seg'], bi, 1e-4, 'bi_seg')
        _close(raw['endp_est'], en, 1e-4, 'endp')
        for k in ('proposal_conf', 'ext2', 'cls2', 'offset2', 'orient'):
            print(k, 'config-5 head output error', _close(raw[k], ref[k], 1e-4, k))
        out = net5(batch)
    assert out['lane_maps']['cls_offset_smooth'][0].shape == (72, 144, 2)


def test_las_file_to_bev_tile(dev, tmp_path):
    """LAS file -> lm_las_decode_points -> lm_bev_raster_batch: the tile equals the C oracle's raster of the same records."""
    from lanemapping_amd import las_io, ops
    from oracle import las_ref, raster_ref
    pts = synth.las_points(77, 300000)
    path = str(tmp_path / 'tile.las')
    off = np.array([351200.0, 3433000.0, 12.0])
    las_ref.write_las(path, pts[:, :3].astype(np.float64) + off, pts[:, 3], point_format=1, offset=tuple(off))
    dev_pts, _ = las_io.read_las_raw(path, dev, shift=off)
    host_pts = las_ref.read_las_ref(path, shift=off, normalise=False).astype(np.float32)
    assert np.array_equal(dev_pts.cpu().numpy(), host_pts)
    par = ops.make_raster_params(local_min_ele=-0.5, ele_reso=0.02)
    out, u8 = ops.bev_raster(dev_pts, par, want_u8=True)
    ref = raster_ref.raster(host_pts, raster_ref.params(local_min_ele=-0.5, ele_reso=0.02))
    assert np.array_equal(u8.cpu().numpy(), ref)


def test_runner_las_to_map_chain(dev, net, tmp_path):
    """Runner.infer_las_to_map: LAS + parameter files -> BEV (GPU) -> polylines -> LAS-frame lines -> merged map.  The 3-D
    lines of every tile equal the oracle chain (numpy LAS reader -> C raster oracle -> reference-pinned img2pc restatement)
    applied to the product's own 2-D polylines."""
    import json
    from lanemapping_amd import io_utils
    from lanemapping_amd.runner import Runner
    from oracle import las_ref, raster_ref, img2pc_ref
    pairs = []
    for t in range(3):
        pts = synth.las_points(900 + t, 250000)
        off = np.array([351200.0 + 40.0 * t, 3433000.0, 12.0])
        quat_trans = [3.0 + 40.0 * t, -2.0, 0.5, 0.999, 0.01, -0.02, 0.03]
        par = raster_ref.params(quat=quat_trans[3:], trans=quat_trans[:3], local_min_ele=-0.5, ele_reso=0.02)
        # tile-frame cloud -> LAS frame: rotate by q, translate, add the read offset (what the param file describes)
        world = np.stack([img2pc_ref.rotate(np.array(quat_trans[3:]), p[:3]) for p in pts[:, :3].astype(np.float64)]) + quat_trans[:3] + off
        las = str(tmp_path / f'18101{t}_0209_a.las')
        las_ref.write_las(las, world, pts[:, 3], point_format=1, offset=tuple(off))
        sp = lambda v: ' '.join(repr(float(x)) for x in v)
        prm = str(tmp_path / f'18101{t}_0209_a.txt')
        with open(prm, 'w') as f:
            f.write('\n'.join(['coor_las_path', las, 'las_read_offset', sp(off), 'las_rotation_trans_quan', sp(quat_trans),
                               'bev_img_offset', '0.0 0.0', 'img_reso', '0.05 0.05', 'local_min_ele', '-0.5', 'ele_reso', '0.02', '']))
        pairs.append((las, prm))
    r = Runner.__new__(Runner)
    r.cfg, r.device, r.net = net.cfg, dev, net
    out = str(tmp_path / 'out')
    lines3d, merged = r.infer_las_to_map(pairs, work_dirs=out, batch_size=2)
    assert len(lines3d) == 3
    for las, prm in pairs:
        name = os.path.basename(las)[0:11]
        params = io_utils.load_pc_2_img_transform_paras(prm)
        host_pts = las_ref.read_las_ref(las, shift=params['las_read_offset'], normalise=False).astype(np.float32)
        q = params['las_rotation_trans_quan']
        tile = raster_ref.raster(host_pts, raster_ref.params(quat=q[3:], trans=q[:3], local_min_ele=-0.5, ele_reso=0.02))
        assert int((tile.sum(axis=2) > 0).sum()) > 100000                       # the cloud really lands on the tile
        seqs, lens, _, _ = io_utils.load_lane_seq(os.path.join(out, name + '.json'))
        want = img2pc_ref.img_to_pc_ref(params, seqs, lens, tile)
        got = json.load(open(os.path.join(out, 'out_pc_seq_json_dir', name + '.json')))
        assert len(got) == len(lens)
        for i, rec in enumerate(got):
            assert np.array_equal(np.asarray(rec['seq']), want[i, :lens[i]])
    assert os.path.exists(os.path.join(out, 'out_pc_seq_json_dir', 'merged.txt')) and len(merged) >= 1
    # the map-level merge (host C++ merger, lm_merge_*) equals the numpy oracle's merge of the same per-tile 3-D files: same arrays
    from oracle import merge_ref
    import glob as _glob
    pc_files = sorted(_glob.glob(os.path.join(out, 'out_pc_seq_json_dir', '*_*.json')) or
                      [f for f in _glob.glob(os.path.join(out, 'out_pc_seq_json_dir', '*.json')) if 'merged' not in f])
    want = merge_ref.merge_lines(pc_files)
    assert len(want) == len(merged)
    for a, b in zip(merged, want):
        assert np.array_equal(a, b)


def test_runner_config4_rowref_json(dev, tmp_path):
    """Runner on the RowRef config (BASELINE configs[3]): PNG tiles -> per-tile JSON through Detector1stage.forward."""
    import json
    from PIL import Image
    from lanemapping_amd.boundary import load_config, build_net_from_config
    from lanemapping_amd.runner import Runner
    net4 = build_net_from_config('Proj28_GFC-T3_RowRef_82_73_laser', device='cpu')
    synth.fill_module_(net4, 2021)
    for t in range(2):
        Image.fromarray(synth.bev_tile_u8(310 + t)).save(str(tmp_path / f'18101{t}_0209_x.png'))
    r = Runner.__new__(Runner)
    r.cfg, r.device, r.net = net4.cfg, dev, net4.to(dev)
    res = r.infer_lane_coordinate_endpoint_semantics(tiles=str(tmp_path), work_dirs=str(tmp_path / 'out'), batch_size=2, write_lane_vertex=True)
    assert len(res) == 2
    for name, (lanes, _) in res.items():
        assert lanes.shape == (72, 144, 2)
        recs = json.load(open(tmp_path / 'out' / (name + '.json')))
        assert len(recs) == int(((lanes[:, :, 0] > 0).sum(axis=1) >= 2).sum())


def test_rowref_config4_tiles_inside_batch8_vs_oracle(dev):
    """Config 4 at the bench batch (B = 8): two tiles INSIDE the batch against the oracle chain run on those tiles alone, margin-aware
    like test_rowref_detector_config4_vs_oracle; the polylines always equal the oracle's line assembly on the product's own decode
    outputs, and the in-batch result equals the product's own single-tile result bit for bit."""
    from lanemapping_amd.boundary import build_net_from_config
    from oracle import net_ref, rowref_ref
    net4 = build_net_from_config('Proj28_GFC-T3_RowRef_82_73_laser', device='cpu')
    synth.fill_module_(net4, 2021)
    sd = {k: v.clone() for k, v in net4.state_dict().items()}
    for c in range(12):
        sd[f'heads.emb_{c}'] = getattr(net4.heads, f'emb_{c}').clone()
    net4 = net4.to(dev)
    seeds = [3100 + i for i in range(8)]
    x = torch.from_numpy(synth.bev_batch(seeds, 1152))
    with torch.no_grad():
        o = net4({'proj': x.to(dev)})
    col_b = net4.heads._col_idx.cpu().numpy()                         # [8,12,144]
    conf_b, cls_b = o['conf'].numpy(), o['cls'].numpy()
    lanes_b = [np.array(l) for l in o['lane_maps']['cls_offset_smooth']]
    total_flips = 0
    for t in (2, 7):
        with torch.no_grad():
            fea = net_ref.vit_forward(sd, net_ref.fpn_forward(sd, x[t:t + 1])[0])
            ref = rowref_ref.rowref_forward(sd, fea)
            o1 = net4({'proj': x[t:t + 1].to(dev)})
        assert np.array_equal(net4.heads._col_idx.cpu().numpy()[0], col_b[t]), 'in-batch tile differs from the single-tile run'
        assert np.array_equal(np.array(o1['lane_maps']['cls_offset_smooth'][0]), lanes_b[t])
        for c in range(12):
            e, p = ref[f'ext2_{c}'][0], ref[f'cls2_{c}'][0]
            want = np.where(e.argmax(dim=1).numpy() == 0, p.argmax(dim=1).numpy(), -1)
            top2 = torch.topk(p, 2, dim=1).values
            margin = torch.minimum((e[:, 0] - e[:, 1]).abs(), top2[:, 0] - top2[:, 1]).numpy()
            bad = np.flatnonzero(col_b[t][c] != want)
            total_flips += bad.size
            assert np.all(margin[bad] < 1e-4), f'tile {t} lane {c}: decision differs from the oracle where its margin is {margin[bad].max():.2e}'
        assert np.array_equal(lanes_b[t], rowref_ref.rowref_pred_lines(conf_b[t], cls_b[t]))
    print(f'config 4, B = 8: {total_flips} decisions flipped inside the oracle margin on 2 tiles')
    assert total_flips <= 16


def test_detector_config5_headline_points_vs_oracle(dev):
    """Config 5 at the bench's point count (4,194,304 points per cloud; the older test uses 1 M): raw head outputs vs the restated
    oracle chain (third-party arithmetic: parity stays unpinned)."""
    from lanemapping_amd.boundary import build_net_from_config
    from oracle import lidar_ref, net_ref
    net5 = build_net_from_config('Proj_polyline_lidarconv_vit_vertex_2', device='cpu')
    synth.fill_module_(net5, 2021)
    sd = {k: v.clone() for k, v in net5.state_dict().items()}
    net5 = net5.to(dev)
    pts = [synth.lidar_points(62, 4194304)]
    pc = dict(net5.cfg.pcencoder)
    pc['gt_downsample_ratio'] = 8
    sd_pc = {k[len('pcencoder.'):]: v for k, v in sd.items() if k.startswith('pcencoder.')}
    with torch.no_grad():
        fea, fea_up, bi, en = lidar_ref.lidar_encoder_ref(pts, sd_pc, pc)
        ref = net_ref.head_forward(sd, net_ref.vit_forward(sd, fea), fea_up)
        raw = net5.forward_raw({'points': [torch.from_numpy(p).to(dev) for p in pts]})
        _close(raw['semantic_seg'], bi, 1e-4, 'bi_seg')
        _close(raw['endp_est'], en, 1e-4, 'endp')
        for k in ('proposal_conf', 'ext2', 'cls2', 'offset2', 'orient'):
            print(k, 'config-5 (4.19 M points) head output error', _close(raw[k], ref[k], 1e-4, k))


# ----------------------------------------------------------------------------------------------- a12: the entry-point contract
def _harness_root(tmp_path, config, n_tiles=5, **subst):
    """A synthetic <data_root> in the reference's layout (cases.write_dataset) + a copy of a repo config pointing at it, named the
    way test_gpu_0.py names the file it loads (logs/<run>/configs_<name>.py)."""
    root = tmp_path / 'data'
    cases.write_dataset(str(root), n_tiles=n_tiles, seed=1901)
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs', config + '.py')).read()
    for a, b in subst.items():
        assert a in src, a
        src = src.replace(a, b)
    src = src.replace("log_dir = './logs'", f"log_dir = {str(tmp_path / 'logs')!r}")
    src = src.replace(src[src.index('dataset_path = '):].split('\n')[0], f'dataset_path = {str(root)!r}')
    path = tmp_path / ('configs_' + config + '.py')
    path.write_text(src)
    return str(root), str(path)


def test_test_gpu_0_body_runs_unchanged(dev, synth_sd, tmp_path, capsys):
    """The body of the reference's test_gpu_0.py:44-62 (config 2 / 3 block), verbatim except for the import line: config file with
    the reference's dataset section and is_gt_avai = True, a DataParallel-style checkpoint, mode_data = cfg.dataset.test,
    mode_view=True, write_lane_vertex=True, eval_coor / eval_semantic on, eval_endp off.  Checks: one JSON per test tile under
    <log_dir>/vis/<dataset type>/<image_name[0:11]>.json, identical to the explicit-tile path; the summed counters equal a
    tile-by-tile recomputation; the nine lines are printed."""
    import json
    from lanemapping_amd import datasets, hostpost, io_utils, metric_utils
    root, path_config = _harness_root(tmp_path, 'Proj_polyline_fpn_vit_vertex_2', **{'is_gt_avai = False': 'is_gt_avai = True'})
    path_ckpt = str(tmp_path / 'best.pth')
    torch.save({'net': {'module.' + k: v for k, v in synth_sd.items()}, 'epoch': 45}, path_ckpt)
    GPUS_EN = '0'
    # ---- test_gpu_0.py:44-62 ----
    from lanemapping_amd.runner import load_config_and_runner          # instead of: from baseline.engine.runner import ...
    cfg, runner = load_config_and_runner(path_config, GPUS_EN)

    cfg.gpus = len(GPUS_EN.split(','))
    print(f'* Config: [{path_config}] is loaded')
    runner.load_ckpt(path_ckpt)
    print(f'* ckpt: [{path_ckpt}] is loaded')
    runner.cfg.show_result = True
    runner.cfg.view_detail = False
    mode_data = cfg.dataset.test        # if infer with evaluation (ground truth available)
    runner.infer_lane_coordinate_endpoint_semantics(path_ckpt=path_ckpt, mode_data=mode_data,  mode_view=True, gt_avail=cfg.is_gt_avai,\
                                                    write_lane_vertex=True, \
                                                    eval_coor=True, eval_endp=False, eval_semantic=True
                                                    )
    # ---- end of the reference's lines ----
    out = capsys.readouterr().out
    for key in ('coordinate_prec', 'coordinate_rec', 'coordinate_f1', 'endpoint_prec', 'endpoint_rec', 'endpoint_f1',
                'semantic_prec', 'semantic_rec', 'semantic_f1'):
        assert f'{key}={runner.metrics[key]}' in out
    assert cfg.work_dirs == str(tmp_path / 'logs') + '/vis/LaserLaneProposal'
    stems = cases.dataset_stems(5)
    written = sorted(os.listdir(cfg.work_dirs))
    assert written == sorted(s[0:11] + '.json' for s in stems)
    # the same tiles through the explicit-tile path (no labels): identical polylines, identical files
    res = runner.infer_lane_coordinate_endpoint_semantics(tiles=os.path.join(root, 'cropped_tiff'), write_lane_vertex=True,
                                                          work_dirs=str(tmp_path / 'explicit'))
    assert runner.metrics['coordinate_f1'] == 0. and not runner.counters.any()
    counters = np.zeros(12)
    ents = {e['stem'][0:11]: e for e in datasets.split_entries(cfg.dataset.test, cfg)}
    for name, (lanes, endp) in res.items():
        assert open(os.path.join(cfg.work_dirs, name + '.json')).read() == open(tmp_path / 'explicit' / (name + '.json')).read()
        assert json.load(open(os.path.join(cfg.work_dirs, name + '.json'))) == io_utils.lane_records(io_utils.pack_lane_vertices(lanes))
        gt = datasets.load_eval_gt(ents[name], cfg)
        counters[0:4] += metric_utils.cal_coor_measures(gt['lc_coor_raw'], lanes[:, :, 0], 'conf', offset_thre=cfg.validate_buffer)[3:7]
        counters[8:12] += metric_utils.eval_metric_line_segmentor(hostpost.raster_semantic_map(lanes), gt['mask'], bi_seg=False,
                                                                  semantics=2, buff=cfg.validate_buffer)[3:7]
    # (second labelled run: eval_endp on as well, nothing written)
    runner.infer_lane_coordinate_endpoint_semantics(mode_data=cfg.dataset.test, gt_avail=True, batch_size=2)
    assert np.array_equal(runner.counters[0:4], counters[0:4]) and np.array_equal(runner.counters[8:12], counters[8:12])
    assert runner.counters[3] > 0 and runner.counters[7] > 0 and runner.counters[11] > 0          # GT vertices / endpoints / pixels were scored
    assert 0. <= runner.metrics['endpoint_f1'] <= 1. and sorted(os.listdir(cfg.work_dirs)) == written


def test_test_gpu_0_body_with_two_gpu_ids(dev, synth_sd, tmp_path, capfd, monkeypatch):
    """GPUS_EN = '0,1' (test_gpu_0.py:7-9; DataParallel(device_ids=range(cfg.gpus)), runner.py:103-104): the same body with two ids.
    load_config_and_runner returns a MultiGpuRunner whose call starts one FRESH process per id (both on this box's GPU through the
    LANEMAP_TEST_DEVICE hook, gloo), shards the test split, gathers once and lets rank 0 write: JSON files byte-identical to the
    one-id run, the same nine printed lines (from rank 0, once), the same counters and the same returned results; then the K-Lane
    and Segmentor entries through the same fan-out."""
    from lanemapping_amd.runner import load_config_and_runner
    from lanemapping_amd.runner_ranks import MultiGpuRunner
    path_ckpt = str(tmp_path / 'best.pth')
    torch.save({'net': {'module.' + k: v for k, v in synth_sd.items()}, 'epoch': 45}, path_ckpt)
    monkeypatch.setenv('LANEMAP_TEST_DEVICE', '0')
    monkeypatch.setenv('LANEMAP_RANKS_TIMEOUT', '900')          # (a hung rank must fail this test, not hang the suite)
    out, runs = {}, {}
    for GPUS_EN in ('0', '0,1'):
        sub = tmp_path / ('ids' + str(len(GPUS_EN.split(','))))
        sub.mkdir()
        root, path_config = _harness_root(sub, 'Proj_polyline_fpn_vit_vertex_2', **{'is_gt_avai = False': 'is_gt_avai = True'})
        cfg, runner = load_config_and_runner(path_config, GPUS_EN)
        cfg.gpus = len(GPUS_EN.split(','))
        runner.load_ckpt(path_ckpt)
        runner.cfg.show_result = True
        runner.cfg.view_detail = False
        mode_data = cfg.dataset.test
        capfd.readouterr()
        res = runner.infer_lane_coordinate_endpoint_semantics(path_ckpt=path_ckpt, mode_data=mode_data, mode_view=True, gt_avail=cfg.is_gt_avai,
                                                              write_lane_vertex=True, eval_coor=True, eval_endp=True, eval_semantic=True)
        out[GPUS_EN] = capfd.readouterr().out
        runs[GPUS_EN] = (cfg, runner, res)
    (cfg1, r1, res1), (cfg2, r2, res2) = runs['0'], runs['0,1']
    assert isinstance(r2, MultiGpuRunner) and not isinstance(r1, MultiGpuRunner) and cfg2.gpus == 2
    names = sorted(os.listdir(cfg1.work_dirs))
    assert len(names) == 5 and sorted(os.listdir(cfg2.work_dirs)) == names
    for n in names:
        assert open(os.path.join(cfg1.work_dirs, n), 'rb').read() == open(os.path.join(cfg2.work_dirs, n), 'rb').read(), n
    assert list(res1) == list(res2)
    for k in res1:
        assert np.array_equal(res1[k][0], res2[k][0]) and np.array_equal(res1[k][1], res2[k][1]), k
    assert np.array_equal(r1.counters, r2.counters) and r1.metrics == r2.metrics and r1.counters[3] > 0 and r1.counters[7] > 0
    lines = [f'{k}={v}' for k, v in r1.metrics.items()]
    for text in (out['0'], out['0,1']):
        assert [l for l in text.splitlines() if l.split('=')[0] in r1.metrics] == lines      # nine lines, once (rank 0 only)
    # a rank that fails (an unknown keyword reaches Runner in the rank processes) ends the call with that rank's message
    with pytest.raises(RuntimeError, match='GPU ranks failed(.|\n)*no_such_keyword'):
        r2._launch('infer_lane_coordinate_endpoint_semantics', {'no_such_keyword': 1})
    # K-Lane (config 4) and Segmentor (config 1) entries, one id vs two
    for config, entry, n_tiles in (('Proj28_GFC-T3_RowRef_82_73_laser', 'infer_lane_coordinate', 3),
                                   ('Proj_FPN_Seg', 'infer_lane_geometry_segmentation_segmentor', 3)):
        got = []
        for GPUS_EN in ('0', '0,1'):
            sub = tmp_path / (config[:8] + str(len(GPUS_EN)))
            sub.mkdir()
            _, path = _harness_root(sub, config, n_tiles=n_tiles)
            cfg, runner = load_config_and_runner(path, GPUS_EN)
            synth.fill_module_(runner.net, 2021)                       # edits of runner.net travel to the ranks
            kw = dict(gt_avail=True, write_lane_vertex=False) if entry == 'infer_lane_coordinate' else {}
            got.append((getattr(runner, entry)(path_ckpt=None, mode_view=True, **kw), runner))
        (ra, a), (rb, b) = got
        assert list(ra) == list(rb) and len(ra) == n_tiles
        for k in ra:
            assert np.array_equal(ra[k][0], rb[k][0]) and ra[k][0].dtype == rb[k][0].dtype and np.array_equal(ra[k][1], rb[k][1]), (config, k)
        assert np.array_equal(a.counters, b.counters) and a.metrics == b.metrics and a.counters[3] > 0, config


def test_klane_and_segmentor_entries(dev, tmp_path, capsys):
    """test_gpu_0.py:66 / :69: `runner.infer_lane_coordinate(path_ckpt=..., mode_view=True, gt_avail=True, write_lane_vertex=False)`
    on the K-Lane RowRef config and `runner.infer_lane_geometry_segmentation_segmentor(path_ckpt=..., mode_view=True)` on the
    Segmentor config, each over cfg.dataset.test of a synthetic LaserLane <data_root>."""
    from lanemapping_amd import datasets, metric_utils
    from lanemapping_amd.runner import load_config_and_runner
    root, path4 = _harness_root(tmp_path, 'Proj28_GFC-T3_RowRef_82_73_laser', n_tiles=3)
    cfg, runner = load_config_and_runner(path4, '0')
    synth.fill_module_(runner.net, 2021)
    path_ckpt = str(tmp_path / 'klane.pth')
    torch.save({'net': {'module.' + k: v for k, v in runner.net.state_dict().items()}}, path_ckpt)
    res = runner.infer_lane_coordinate(path_ckpt=path_ckpt, mode_view=True, gt_avail=True, write_lane_vertex=False)
    assert sorted(res) == sorted(s[0:11] for s in cases.dataset_stems(3)) and os.listdir(cfg.work_dirs) == []
    assert cfg.work_dirs.endswith('/vis/LaserLane')
    tot = np.zeros(4)
    for e in datasets.split_entries(cfg.dataset.test, cfg):
        gt = datasets.load_eval_gt(e, cfg, merge_connect_lines=False)
        tot += metric_utils.cal_coor_measures(datasets.klane_coor_label(gt['label_raw'], 12), res[e['stem'][0:11]][0][:12, :, 0], 'conf',
                                              offset_thre=cfg.validate_buffer)[3:7]
    assert np.array_equal(runner.counters[0:4], tot) and tot[3] > 0
    assert f"coordinate_f1={runner.metrics['coordinate_f1']}" in capsys.readouterr().out
    _, path1 = _harness_root(tmp_path / 'seg', 'Proj_FPN_Seg', n_tiles=2)
    cfg1, runner1 = load_config_and_runner(path1, '0')
    synth.fill_module_(runner1.net, 2021)
    res1 = runner1.infer_lane_geometry_segmentation_segmentor(path_ckpt=None, mode_view=True)
    assert len(res1) == 2 and all(v[0].shape == (1152, 1152) for v in res1.values())
    assert runner1.counters[3] > 0 and runner1.counters[7] > 0 and 'sem_conf_f1=' in capsys.readouterr().out
