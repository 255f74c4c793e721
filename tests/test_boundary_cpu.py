"""CPU: the drop-in boundary (registries, config loader, state-dict layout), the C-ABI library surface, and the
host-side C++ tail (endpoint clustering + polyline assembly) against the reference-generated goldens."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

import cases
from lanemapping_amd import hostpost, synth
from lanemapping_amd._lib import lib, SIGNATURES, LanemapHipError

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'lanemap_hip.h')).read()
    declared = set(re.findall(r'\b(lm_[a-z0-9_]+)\s*\(', header))
    L = lib()
    assert declared == set(SIGNATURES), (declared ^ set(SIGNATURES))
    for name in declared:
        assert hasattr(L, name), name
    assert L.lm_abi_version() == 1


def test_c_abi_reports_errors_without_a_gpu():
    L = lib()
    rc = L.lm_conv2d_nhwc_mfma_f32(None, None, 64, None, 128, None, None, None, 0, 0, None, 64,
                                   1, 8, 8, 64, 64, 3, 3, 1, 1, 1, 1, 0)
    assert rc == 1 and b'null pointer' in L.lm_last_error()


def test_registry_semantics():
    from lanemapping_amd.registry import Registry, build_from_cfg, NET, HEADS
    from lanemapping_amd import boundary  # noqa: F401
    assert {'Detector1stage', 'Segmentor'} <= set(NET.module_dict)
    assert 'ColumnProposal2' in HEADS.module_dict
    r = Registry('x')

    @r.register_module
    class A:
        def __init__(self, a=1, cfg=None):
            self.a, self.cfg = a, cfg
    with pytest.raises(KeyError):
        r.register_module(A)
    with pytest.raises(TypeError):
        r.register_module(3)
    obj = build_from_cfg({'type': 'A', 'a': 5}, r, default_args={'cfg': 'c'})
    assert (obj.a, obj.cfg) == (5, 'c')
    with pytest.raises(KeyError):
        build_from_cfg({'type': 'Missing'}, r)
    with pytest.raises(TypeError):
        build_from_cfg({'type': 3}, r)


def test_config_loader(tmp_path):
    from lanemapping_amd.boundary import load_config
    cfg = load_config('Proj_polyline_fpn_vit_vertex_2')
    assert cfg.heads.type == 'ColumnProposal2' and cfg.heads.num_prop == 72 and cfg.backbone.depth == 3
    assert cfg.pcencoder.pretrained is False and cfg.vit_seg is True
    with pytest.raises(AttributeError):
        cfg.no_such_key
    p = tmp_path / 'c.py'
    p.write_text("import os\nnet = dict(type='Segmentor')\npcencoder = dict(type='PostProjector2', pretrained=True)\nseg_thre = 0.3\n")
    c2 = load_config(str(p))
    assert c2.seg_thre == 0.3 and c2.is_gt_avai is False and 'os' not in c2 and c2.pcencoder.pretrained is False


def test_state_dict_layout_matches_reference_checkpoints(synth_sd, tmp_path):
    """600 entries for config 2, incl. dead parameters; DataParallel 'module.' prefix accepted."""
    from lanemapping_amd.boundary import build_net_from_config, load_reference_checkpoint
    assert len(synth_sd) == 600
    for k, shape in {'conv1.weight': (144, 144, 3, 3), 'pcencoder.fpn.model_buttomup.layer3.5.bn2.running_var': (256,),
                     'pcencoder.fpn.layer3.0.downsample.0.weight': (256, 128, 1, 1), 'backbone.pos_embedding': (1, 324, 512),
                     'backbone.transformer.layers.2.1.fn.net.3.weight': (512, 2048), 'heads.emb_71': (512,),
                     'heads.proposal_confidence.1.weight': (2, 23040), 'heads.endpoint.0.weight': (4, 17, 3, 3),
                     'heads.generate_line_proposal.0.layers.1.2.weight': (16, 8, 3, 3)}.items():
        assert tuple(synth_sd[k].shape) == shape, k
    path = tmp_path / 'best.pth'
    torch.save({'net': {'module.' + k: v for k, v in synth_sd.items()}, 'epoch': 3}, path)
    net = build_net_from_config('Proj_polyline_fpn_vit_vertex_2', device='cpu')
    res = load_reference_checkpoint(net, str(path), strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert torch.equal(net.state_dict()['heads.ext2.0.weight'], synth_sd['heads.ext2.0.weight'])


def test_product_path_fails_loudly_without_gpu():
    from lanemapping_amd.boundary import build_net_from_config
    net = build_net_from_config('Proj_FPN_Seg', device='cpu')
    with pytest.raises((LanemapHipError, RuntimeError, AssertionError)):
        net({'proj': torch.zeros(1, 3, 64, 64)})


@pytest.mark.parametrize('i', range(cases.NUM_POSTPROC_CASES))
def test_cpp_polyline_assembly_vs_reference_golden(golden, i):
    g = golden('g6_postproc.npz')
    c = cases.postproc_case(i)
    pc = np.stack([1 - c['prop_conf1'], c['prop_conf1']], 1).astype(np.float32)
    V, E = hostpost.assemble_polylines(pc, c['prop_v_ext'].astype(np.float32), c['cls_offset'], c['bi_seg_rows'], c['endp_pts'])
    E = E[np.lexsort((E[:, 1], E[:, 0]))] if len(E) else E.reshape(0, 2)
    assert np.array_equal(V, g[f'V{i}'])
    assert np.array_equal(E, g[f'E{i}'].reshape(-1, 2))


def test_cpp_assembly_on_reference_e2e_decode(golden):
    g = golden('g10_e2e.npz')
    V, E = hostpost.assemble_polylines(g['prop_conf'][0], g['prop_v_ext'][0].astype(np.float32), g['cls_offset'][0],
                                       g['bi_seg_rows'][0], g['endp'])
    assert np.array_equal(V, g['cls_offset_smooth'])
    assert np.array_equal(E[np.lexsort((E[:, 1], E[:, 0]))], g['endp_final'])


def test_postproc_is_chaotic_on_g10(golden):
    """Documents why end-to-end vertex parity is pinned stage-wise: 1e-7 of input noise already changes the
    polyline set on the random-weight tile (the reference's algorithm, not an implementation artefact)."""
    g = golden('g10_e2e.npz')
    rng = np.random.default_rng(0)
    V, _ = hostpost.assemble_polylines(g['prop_conf'][0], g['prop_v_ext'][0].astype(np.float32),
                                       g['cls_offset'][0] + 1e-7 * rng.standard_normal(g['cls_offset'][0].shape),
                                       g['bi_seg_rows'][0], g['endp'])
    assert np.abs(V - g['cls_offset_smooth']).max() > 1.0


def test_cpp_endpoint_clustering_vs_oracle(golden):
    from oracle import decode_ref
    g = golden('g5_decode.npz')
    raw = cases.decode_inputs(int(g['input_seed']), batch=2)
    for b in range(2):
        _, K, order = decode_ref.select_endpoints(raw['endp_est'][b, 0], k0=240)
        full = decode_ref.select_endpoints(raw['endp_est'][b, 0], k0=240)[0]
        sig = torch.sigmoid(torch.from_numpy(raw['endp_est'][b, 0, 20:-20, 20:-20])).numpy().reshape(-1)
        top = np.lexsort((np.arange(sig.size), -sig.astype(np.float64)))[:512].astype(np.int32)
        pts, k_used = hostpost.cluster_endpoints(top)
        assert k_used == K
        assert np.array_equal(pts[np.lexsort((pts[:, 1], pts[:, 0]))], g[f'endp{b}'])
        assert np.array_equal(np.stack(np.nonzero(full), 1), g[f'endp{b}'])
    with pytest.raises(LanemapHipError):
        hostpost.cluster_endpoints(np.arange(10, dtype=np.int32))       # fewer scores than K=240 -> loud error


def test_semantic_raster_vs_oracle(golden):
    from oracle import postproc_ref
    g = golden('g6_postproc.npz')
    for i in (0, 2, 12):
        assert np.array_equal(hostpost.raster_semantic_map(g[f'V{i}']), postproc_ref.raster_semantic_map(g[f'V{i}']))


def test_empty_and_degenerate_tiles():
    z = np.zeros((72, 144), dtype=np.float32)
    V, E = hostpost.assemble_polylines(np.zeros((72, 2), np.float32), z, z.astype(np.float64), np.zeros((144, 1152), np.float32),
                                       np.zeros((0, 2), np.int32))
    assert (V[:, :, 0] == -1).all() and (V[:, :, 1] == 0).all() and len(E) == 0
    # every proposal firing on every row at the same column: must terminate and give a single line
    pc = np.tile(np.array([[0.1, 0.9]], np.float32), (72, 1))
    V, _ = hostpost.assemble_polylines(pc, np.ones((72, 144), np.float32), np.full((72, 144), 70.0), np.ones((144, 1152), np.float32),
                                       np.array([[600, 560]], np.int32))
    assert ((V[:, :, 0] > 0).sum(1) >= 2).sum() == 1


def test_json_output_is_byte_identical_to_reference(golden, tmp_path):
    from lanemapping_amd import io_utils
    g = golden('g6_postproc.npz')
    p = tmp_path / 'tile.json'
    io_utils.save_lane_seq_2d(io_utils.pack_lane_vertices(g['V0']), str(p))
    assert p.read_text() == open(os.path.join(ROOT, 'tests', 'golden', 'g9_lanes.json')).read()
    seq, lens, init, end = io_utils.load_lane_seq(str(p), dim_coor=3)
    assert seq.shape[0] == len(lens) == 4 and seq.shape[2] == 3
    t = tmp_path / 'tile.txt'
    io_utils.save_lane_seq_2d(io_utils.pack_lane_vertices(g['V0']), str(t))
    first = t.read_text().splitlines()[0].split(' ')
    assert len(first) == 4 and first[-1] == '0'


def test_param_file_parser(tmp_path):
    from lanemapping_amd import io_utils
    p = tmp_path / '181013_0130.txt'
    p.write_text('las path:\n/data/a.las\nlas read offset:\n1000.5 2000.25 10\nrotation translation quaternion:\n'
                 '1 2 3 0.7071068 0 0 0.7071068\nbev image offset:\n-28.8 -28.8\nimage resolution:\n0.05 0.05\n'
                 'local min elevation:\n-2.5\nelevation resolution:\n0.02\n')
    d = io_utils.load_pc_2_img_transform_paras(str(p))
    assert d['coor_las_path'] == '/data/a.las' and d['las_read_offset'] == [1000.5, 2000.25, 10.0]
    assert d['las_rotation_trans_quan'][3:] == [0.7071068, 0.0, 0.0, 0.7071068] and d['ele_reso'] == 0.02
    rp = io_utils.raster_params_from_file(str(p))
    assert abs(rp.quat[0] - 0.7071068) < 1e-7 and list(rp.trans) == [1.0, 2.0, 3.0] and abs(rp.local_min_ele + 2.5) < 1e-7


def test_cpp_trace_lines_vs_rowref_golden(golden):
    from oracle import rowref_ref
    g = golden('g8_rowref.npz')
    for b in range(2):
        conf_pred = np.where(g['conf'][b] > 0.5, 1, 0)
        cls_idx = np.argmax(torch.softmax(torch.from_numpy(g['cls'][b].astype(np.float64)), 0).numpy(), 0)
        cls_idx[cls_idx == 12] = 255
        cls_idx[conf_pred == 0] = 255
        lines = np.zeros((12, 144)) - 1.0
        for c in range(12):
            r, w = np.nonzero(cls_idx == c)
            lines[c, r] = w / 144 * 1152. + 4
        assert np.array_equal(hostpost.trace_lines(lines), g['pred_lines'][b])


def test_rowref_state_dict_layout():
    from lanemapping_amd.boundary import build_net_from_config
    net = build_net_from_config('Proj28_GFC-T3_RowRef_82_73_laser', device='cpu')
    sd = net.state_dict()
    assert tuple(sd['heads.cls2_11.2.weight'].shape) == (144, 512, 1) and tuple(sd['heads.to_token.1.weight'].shape) == (1024, 5760)
    assert tuple(sd['heads.tr_lane_correlator.2.weight'].shape) == (5760, 1024)
    assert not any('emb_' in k for k in sd)          # as on a real GPU in the reference: emb_c never reach a checkpoint
    assert len([k for k in sd if k.startswith('heads.')]) == 449


# ----------------------------------------------------------------------------------------------- config 5 (LiDAR encoder)
def test_lidar_encoder_state_dict_layout():
    """mmdet3d SparseEncoder naming + mmcv.ops spconv weight layout [kD,kH,kW,Cin,Cout] under the reference's keys."""
    from lanemapping_amd.boundary import build_net_from_config
    net5 = build_net_from_config('Proj_polyline_lidarconv_vit_vertex_2', device='cpu')
    sd = net5.state_dict()
    bb = 'pcencoder.lidar_modal_extractor.backbone.'
    want = {
        bb + 'conv_input.0.weight': (3, 3, 3, 4, 16), bb + 'conv_input.1.running_var': (16,),
        bb + 'encoder_layers.encoder_layer1.0.conv1.weight': (3, 3, 3, 16, 16),
        bb + 'encoder_layers.encoder_layer1.1.bn2.bias': (16,),
        bb + 'encoder_layers.encoder_layer1.2.0.weight': (3, 3, 3, 16, 32),
        bb + 'encoder_layers.encoder_layer3.2.0.weight': (3, 3, 3, 64, 128),
        bb + 'encoder_layers.encoder_layer4.1.conv2.weight': (3, 3, 3, 128, 128),
        bb + 'conv_out.0.weight': (3, 1, 1, 128, 128), bb + 'conv_out.1.weight': (128,),
        'pcencoder.fea_aligner.0.weight': (64, 128, 3, 3), 'pcencoder.fea_conv.0.bias': (64,),
        'pcencoder.output_layer_binary_seg.weight': (3, 64, 1, 1), 'pcencoder.output_layer_endp.bias': (1,),
    }
    for k, shp in want.items():
        assert k in sd and tuple(sd[k].shape) == shp, k
    assert not any('encoder_layer4.2' in k for k in sd)            # last stage has no down-sampling conv
    assert not any(k.endswith('.0.bias') and 'lidar_modal_extractor' in k for k in sd)   # spconv layers carry no bias
    vox = net5.pcencoder.lidar_modal_extractor['voxelize']
    assert vox.grid_xyz == [575, 575, 9] and abs(vox.voxel_size[0] - 30.0 / 575) < 1e-7
    net5.load_state_dict(sd, strict=True)


def test_voxelize_ref_known_answers():
    from oracle import lidar_ref
    lo, vs, grid = [0., 0., 0.], [1., 1., 1.], [4, 4, 2]
    pts = np.array([[2.5, 1.5, 0.5, 1.0],     # voxel A (z0,y1,x2)  first
                    [0.5, 0.5, 1.5, 2.0],     # voxel B (1,0,0)
                    [2.2, 1.1, 0.9, 3.0],     # A
                    [4.0, 1.0, 0.0, 9.0],     # x == grid edge -> dropped
                    [-0.1, 1.0, 0.0, 9.0],    # below range -> dropped
                    [2.9, 1.9, 0.1, 5.0],     # A (third point: dropped by max_points=2)
                    [3.5, 3.5, 1.5, 4.0]],    # voxel C (1,3,3)
                   np.float32)
    f, c = lidar_ref.voxelize_ref([pts, pts[:2]], lo, vs, grid, max_points=2, max_voxels=10)
    assert c.tolist() == [[0, 0, 1, 2], [0, 1, 0, 0], [0, 1, 3, 3], [1, 0, 1, 2], [1, 1, 0, 0]]
    assert np.allclose(f[0], (pts[0] + pts[2]) / 2) and np.allclose(f[1], pts[1]) and np.allclose(f[2], pts[6])
    f2, c2 = lidar_ref.voxelize_ref([pts], lo, vs, grid, max_points=10, max_voxels=2)
    assert c2.tolist() == [[0, 0, 1, 2], [0, 1, 0, 0]]             # first-appearance numbering, capped
    assert np.allclose(f2[0], (pts[0] + pts[2] + pts[5]) / 3)
    f3, c3 = lidar_ref.voxelize_ref([np.zeros((0, 4), np.float32)], lo, vs, grid, 2, 10)
    assert f3.shape == (0, 4) and c3.shape == (0, 4)


def test_sparse_encoder_ref_against_sitewise_definition():
    """The oracle evaluates spconv layers as masked dense conv3d; check that against the site-wise (rulebook) definition
    on a tiny active set: SubMConv3d keeps the set, SparseConv3d activates every window holding an active input."""
    import torch
    from oracle import lidar_ref
    rng = np.random.RandomState(3)
    D, H, W, cin, cout = 5, 6, 7, 3, 4
    act = rng.rand(D, H, W) > 0.7
    zs, ys, xs = np.nonzero(act)
    feats = rng.randn(len(zs), cin).astype(np.float32)
    w = rng.randn(3, 3, 3, cin, cout).astype(np.float32)
    x = torch.zeros(1, cin, D, H, W)
    m = torch.zeros(1, 1, D, H, W)
    x[0, :, zs, ys, xs] = torch.from_numpy(feats).t()
    m[0, 0, zs, ys, xs] = 1
    sd = {'w': torch.from_numpy(w)}
    site = {(z, y, x_): i for i, (z, y, x_) in enumerate(zip(zs, ys, xs))}
    got = lidar_ref._subm(x, m, sd, 'w')[0]
    for (z, y, x_), i in site.items():
        acc = np.zeros(cout)
        for kz in range(3):
            for ky in range(3):
                for kx in range(3):
                    j = site.get((z - 1 + kz, y - 1 + ky, x_ - 1 + kx))
                    if j is not None:
                        acc += feats[j].astype(np.float64) @ w[kz, ky, kx]
        assert np.allclose(got[:, z, y, x_].numpy(), acc, atol=1e-5)
    assert float((got * (1 - m[0])).abs().max()) == 0.0
    pad, stride = (1, 1, 0), (2, 2, 2)
    got2, m2 = lidar_ref._spconv(x, m, sd, 'w', (3, 3, 3), stride, pad)
    Do, Ho, Wo = got2.shape[2:]
    for oz in range(Do):
        for oy in range(Ho):
            for ox in range(Wo):
                acc, hit = np.zeros(cout), False
                for kz in range(3):
                    for ky in range(3):
                        for kx in range(3):
                            j = site.get((oz * 2 - pad[0] + kz, oy * 2 - pad[1] + ky, ox * 2 - pad[2] + kx))
                            if j is not None:
                                hit = True
                                acc += feats[j].astype(np.float64) @ w[kz, ky, kx]
                assert bool(m2[0, 0, oz, oy, ox]) == hit
                assert np.allclose(got2[0, :, oz, oy, ox].numpy(), acc if hit else 0, atol=1e-5)


# ----------------------------------------------------------------------------------------------- f1: BEV polylines -> LAS frame
def test_polyline_backproject_golden_g12(golden, tmp_path):
    """lm_polyline_backproject (host C++) vs the reference's transform_coordinate_from_img_2_pc: bit-identical float64,
    and the per-tile file driver reproduces the reference's JSON / TXT text."""
    import cases
    from lanemapping_amd import coor_img2pc
    g = golden('g12_img2pc.npz')
    for i, seed in enumerate(g['seeds']):
        params, seqs, lens, tile = cases.img2pc_case(int(seed))
        keep = tile.copy()
        out = coor_img2pc.transform_coordinate_from_img_2_pc(params, seqs.copy(), lens, tile)
        assert np.array_equal(out, g[f'out_{i}'])
        assert np.array_equal(tile, keep)                       # the caller's tile is not modified
    cases.write_img2pc_files(str(tmp_path), 501)
    coor_img2pc.transform_coordinate_from_img_2_pc_single(f'{tmp_path}/t.json', f'{tmp_path}/t.png', f'{tmp_path}/t.txt',
                                                          f'{tmp_path}/o.json', f'{tmp_path}/o.txt')
    assert open(f'{tmp_path}/o.json').read() == str(g['file_json'])
    assert open(f'{tmp_path}/o.txt').read() == str(g['file_txt'])


def test_polyline_backproject_errors_and_roundtrip():
    """Error behaviour + LAS -> BEV -> LAS closure against the rasteriser's C oracle: every vertex placed on an occupied
    pixel maps back to within one pixel pitch / one elevation step of a point that landed in that pixel."""
    from lanemapping_amd import coor_img2pc, synth
    from lanemapping_amd._lib import LanemapHipError
    from oracle import raster_ref
    params = {'img_reso': [0.05, 0.05], 'bev_img_offset': [0.0, 0.0], 'ele_reso': 0.02, 'local_min_ele': -0.5,
              'las_read_offset': [0.0, 0.0, 0.0], 'las_rotation_trans_quan': [1.5, -2.0, 0.25, 0.9, 0.02, -0.03, 0.4]}
    with pytest.raises(LanemapHipError):                         # an all-empty tile has no elevation to borrow
        coor_img2pc.transform_coordinate_from_img_2_pc(params, np.array([[[5., 5.], [13., 6.]]]), [2], np.zeros((32, 32, 3), np.uint8))
    with pytest.raises(LanemapHipError):                         # vertex outside the tile
        coor_img2pc.transform_coordinate_from_img_2_pc(params, np.array([[[5., 40.], [13., 6.]]]), [2], np.ones((32, 32, 3), np.uint8))
    H = W = 128
    q = np.array(params['las_rotation_trans_quan'][3:])
    pts_tile = synth.las_points(7, 60000, extent=H * 0.05)
    # move the tile-frame points into the LAS frame with the reference's forward transform (rotate, translate)
    from oracle import img2pc_ref
    las = np.stack([img2pc_ref.rotate(q, p[:3]) + np.array(params['las_rotation_trans_quan'][:3]) for p in pts_tile[:4000]])
    pts = np.concatenate([las, pts_tile[:4000, 3:4]], axis=1).astype(np.float32)
    rp = raster_ref.params(quat=q, trans=params['las_rotation_trans_quan'][:3], bev_img_offset=(0, 0), img_reso=(0.05, 0.05),
                                local_min_ele=-0.5, ele_reso=0.02)
    img = raster_ref.raster(pts, rp, H, W)
    occ = np.argwhere(img.sum(axis=2) > 1)[:80]
    seqs = occ[None].astype(np.float64)
    out = coor_img2pc.transform_coordinate_from_img_2_pc(params, seqs, [len(occ)], img)
    d = np.sqrt(((out[0][:, None, :2] - las[None, :, :2]) ** 2).sum(-1)).min(axis=1)
    assert float(d.max()) <= 0.05 * 1.5


# ----------------------------------------------------------------------------------------------- f4: LAS header (host part)
def test_las_header_parse(tmp_path):
    from lanemapping_amd import las_io
    from lanemapping_amd._lib import LanemapHipError
    from oracle import las_ref
    rng = np.random.RandomState(5)
    xyz = rng.rand(1000, 3) * [50, 50, 3] + [351200.0, 3433000.0, 10.0]
    inten = rng.randint(0, 65535, 1000)
    p12 = str(tmp_path / 'a.las')
    las_ref.write_las(p12, xyz, inten, point_format=1, version=(1, 2), offset=(351200.0, 3433000.0, 0.0), vlr_bytes=54)
    h = las_io.parse_header(open(p12, 'rb').read())
    assert h['version'] == (1, 2) and h['point_format'] == 1 and h['record_len'] == 28 and h['n_points'] == 1000
    assert h['offset_to_points'] == 227 + 54 and h['scale'] == [0.001] * 3 and h['offset'] == [351200.0, 3433000.0, 0.0]
    p14 = str(tmp_path / 'b.las')
    las_ref.write_las(p14, xyz, inten, point_format=6, version=(1, 4), extra_bytes=3, offset=(351200.0, 3433000.0, 0.0))
    h = las_io.parse_header(np.fromfile(p14, dtype=np.uint8))
    assert h['version'] == (1, 4) and h['record_len'] == 33 and h['n_points'] == 1000 and h['offset_to_points'] == 375
    raw = bytearray(open(p12, 'rb').read())
    with pytest.raises(LanemapHipError):
        las_io.parse_header(b'LAZY' + bytes(raw[4:]))
    with pytest.raises(LanemapHipError):
        las_io.parse_header(bytes(raw[:5000]))                  # truncated point data
    raw[104] |= 0x80                                            # LAZ-compressed marker
    with pytest.raises(LanemapHipError):
        las_io.parse_header(bytes(raw))


# ----------------------------------------------------------------------------------------------- f2: cross-tile merge
@pytest.mark.parametrize('mode,shape', [('RGB', (97, 131, 3)), ('RGBA', (64, 50, 4)), ('L', (33, 77)), ('LA', (40, 41, 2)),
                                        ('RGB', (1, 1, 3)), ('RGB', (3, 1152, 3))])
def test_png_reader_matches_pil(mode, shape):
    """png_io (csrc/png_reader.cpp, host code) == np.array(Image.open()) of the reference's load_img, for every scanline filter."""
    import io
    from PIL import Image
    from lanemapping_amd import png_io
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, shape, dtype=np.uint8)
    ramp = (np.add.outer(np.arange(shape[0]) * 3, np.arange(shape[1]) * 2) % 251).astype(np.uint8)
    smooth = noise.copy()
    if smooth.ndim == 3:
        smooth[..., 0] = ramp
        smooth[..., 1] = ramp // 2
    else:
        smooth = ramp
    seen = set()
    for arr in (noise, smooth, np.zeros(shape, np.uint8)):
        for level in (1, 6, 9):
            buf = io.BytesIO()
            Image.fromarray(arr, mode).save(buf, 'PNG', compress_level=level)
            data = buf.getvalue()
            ref = np.array(Image.open(io.BytesIO(data)))
            got = png_io.decode_png(data)
            assert got.dtype == np.uint8 and got.shape == ref.shape and np.array_equal(got, ref)
            assert png_io.png_info(data) == (shape[0], shape[1], 1 if len(shape) == 2 else shape[2])
            import struct
            import zlib
            pos, z = 8, b''
            while pos < len(data):
                n, t = struct.unpack('>I', data[pos:pos + 4])[0], data[pos + 4:pos + 8]
                z += data[pos + 8:pos + 8 + n] if t == b'IDAT' else b''
                pos += 12 + n
            raw = zlib.decompress(z)
            st = shape[1] * (1 if len(shape) == 2 else shape[2]) + 1
            seen |= {raw[i * st] for i in range(shape[0])}
    if shape[0] > 30:
        assert {1, 2, 4} <= seen or {0, 1, 2} <= seen, seen        # the encoder really exercised several filter types


def test_png_reader_hand_filtered_rows_and_errors(tmp_path):
    """Every filter type incl. Average (which PIL's encoder rarely picks), multi-IDAT streams, and the refusals."""
    import struct
    import zlib
    from lanemapping_amd import png_io
    from lanemapping_amd._lib import LanemapHipError
    rng = np.random.default_rng(6)
    H, W, Cn = 10, 23, 3
    img = rng.integers(0, 256, (H, W, Cn), dtype=np.uint8)
    stride = W * Cn

    def paeth(a, b, c):
        p = a + b - c
        pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
        return a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)

    raw = bytearray()
    for y in range(H):
        ft = y % 5
        cur = img[y].reshape(-1).astype(int)
        up = img[y - 1].reshape(-1).astype(int) if y else np.zeros(stride, int)
        raw.append(ft)
        for i in range(stride):
            a = cur[i - Cn] if i >= Cn else 0
            b, c = up[i], (up[i - Cn] if i >= Cn else 0)
            pred = [0, a, b, (a + b) >> 1, paeth(a, b, c)][ft]
            raw.append((cur[i] - pred) & 255)

    def chunk(t, d):
        return struct.pack('>I', len(d)) + t + d + struct.pack('>I', zlib.crc32(t + d) & 0xffffffff)

    def png(depth=8, color=2, interlace=0, idat_split=3, z=None, w=W, h=H):
        z = zlib.compress(bytes(raw)) if z is None else z
        k = max(1, len(z) // idat_split)
        idats = b''.join(chunk(b'IDAT', z[i:i + k]) for i in range(0, len(z), k))
        return (b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, depth, color, 0, 0, interlace)) + chunk(b'tEXt', b'k\0v')
                + idats + chunk(b'IEND', b''))

    good = png()
    assert np.array_equal(png_io.decode_png(good), img)
    assert np.array_equal(png_io.decode_png(png(idat_split=1)), img)

    # the vector paths of the unfilter step (Sub as a prefix sum, Average / Paeth one pixel per step, TWO consecutive Paeth rows as a wavefront):
    # 3- and 4-byte pixels, widths from one pixel up (every tail length of the 16-byte groups), runs of equal filter types of every length
    def filtered(im, fts):
        hh, ww, cn = im.shape
        st = ww * cn
        out = bytearray()
        for y in range(hh):
            ft = int(fts[y])
            cur = im[y].reshape(-1).astype(int)
            up = im[y - 1].reshape(-1).astype(int) if y else np.zeros(st, int)
            out.append(ft)
            for i in range(st):
                a = cur[i - cn] if i >= cn else 0
                b, c = up[i], (up[i - cn] if i >= cn else 0)
                out.append((cur[i] - [0, a, b, (a + b) >> 1, paeth(a, b, c)][ft]) & 255)
        return bytes(out)

    for cn, color in ((3, 2), (4, 6), (1, 0), (2, 4)):
        for ww in (1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 16, 17, 21, 33):
            hh = 11
            im = rng.integers(0, 256, (hh, ww, cn), dtype=np.uint8)
            im[:, :, 0] = (np.add.outer(np.arange(hh) * 7, np.arange(ww) * 5) % 256).astype(np.uint8)      # (a smooth channel: Paeth picks all three)
            for fts in ([4] * hh, [1] * hh, [3] * hh, [4, 4, 1, 4, 4, 4, 3, 4, 4, 2, 4], list(rng.integers(0, 5, hh))):
                z = zlib.compress(filtered(im, fts), int(rng.integers(0, 10)))
                data = (b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', ww, hh, 8, color, 0, 0, 0)) + chunk(b'IDAT', z) + chunk(b'IEND', b''))
                got = png_io.decode_png(data)
                assert np.array_equal(got.reshape(hh, ww, cn), im), (cn, ww, fts)
    (tmp_path / 'a.png').write_bytes(good)
    (tmp_path / 'b.png').write_bytes(png(idat_split=7))
    batch = png_io.read_png_batch([tmp_path / 'a.png', tmp_path / 'b.png'], threads=2)
    assert batch.shape == (2, H, W, Cn) and np.array_equal(batch[0], img) and np.array_equal(batch[1], img)
    assert np.array_equal(png_io.read_png(tmp_path / 'a.png'), img)

    bad_crc = bytearray(good)
    bad_crc[60] ^= 1
    cases = {'signature': b'JUNK' + good[4:], 'CRC': bytes(bad_crc), '8-bit': png(depth=16), 'palette': png(color=3),
             'interlaced': png(interlace=1), 'truncated': good[:len(good) - 20], 'shorter': png(h=H + 1),
             'larger': png(h=H - 1), 'corrupt': png(z=zlib.compress(bytes(raw))[:-6] + b'\0\0\0\0\0\0')}
    for word, data in cases.items():
        with pytest.raises(LanemapHipError, match=word):
            png_io.decode_png(data)
    (tmp_path / 'c.png').write_bytes(png(w=W + 1, z=zlib.compress(bytes(H * ((W + 1) * Cn + 1)))))
    with pytest.raises(LanemapHipError, match='c.png.*geometry'):
        png_io.read_png_batch([tmp_path / 'a.png', tmp_path / 'c.png'], threads=2)
    with pytest.raises(LanemapHipError, match='cannot open'):
        png_io.read_png_batch([tmp_path / 'a.png', tmp_path / 'missing.png'], threads=1)


def test_png_reader_survives_damaged_files():
    """Untrusted input: random byte flips / truncations (with and without repaired chunk CRCs, so that the zlib and unfilter paths are
    reached) must end in an array of the declared shape or a LanemapHipError - never a crash, a hang or an out-of-bounds write."""
    import io
    import struct
    import zlib
    from PIL import Image
    from lanemapping_amd import png_io
    from lanemapping_amd._lib import LanemapHipError
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, (48, 61, 3), dtype=np.uint8)
    img[:, :, 0] = (np.add.outer(np.arange(48), np.arange(61)) % 251).astype(np.uint8)
    buf = io.BytesIO()
    Image.fromarray(img).save(buf, 'PNG')
    good = buf.getvalue()
    assert np.array_equal(png_io.decode_png(good), img)

    def fix_crcs(data):
        out, pos = bytearray(data[:8]), 8
        while pos + 12 <= len(data):
            n = struct.unpack('>I', data[pos:pos + 4])[0]
            if pos + 12 + n > len(data):
                break
            body = data[pos + 4:pos + 8 + n]
            out += data[pos:pos + 4] + body + struct.pack('>I', zlib.crc32(body) & 0xffffffff)
            pos += 12 + n
        return bytes(out + data[pos:])

    ok = bad = 0
    for trial in range(400):
        data = bytearray(good)
        kind = trial % 4
        if kind == 0:                                           # truncation
            data = data[:int(rng.integers(0, len(data)))]
        else:
            for _ in range(int(rng.integers(1, 4))):
                data[int(rng.integers(8 if kind == 3 else 0, len(data)))] ^= int(rng.integers(1, 256))
        data = bytes(data)
        if kind >= 2:
            data = fix_crcs(data)
        try:
            out = png_io.decode_png(data)
            assert out.dtype == np.uint8 and out.ndim in (2, 3)
            ok += 1
        except LanemapHipError:
            bad += 1
    assert bad > 300 and ok + bad == 400


def test_native_lane_json_is_json_dump_byte_for_byte(tmp_path):
    """csrc/lane_json.cpp == json.dump(lane_records(..), indent=4): random bit patterns, the repr() format switches, empty files."""
    import json
    from lanemapping_amd import io_utils
    rng = np.random.default_rng(1)
    special = [1e16, 9999999999999998.0, 1e-5, 0.0001, 0.00012345, 123456789012345680.0, 5e-324, 1.7976931348623157e308, 1e22,
               2.0 ** -24, 2.0 ** -1074, 2.0 ** 60, 0.1 + 0.2, 1 / 3, 100.0, 1152.0, 3.0, 1e15, 123456789.123, 0.5, 2.5e-7, 1e21, 1e-4,
               9.999999999999999e-05, 2 ** 53 + 2.0, float('inf')]
    for trial in range(120):
        L = int(rng.integers(0, 7))
        v = np.zeros((L, 40, 3))
        kind = trial % 3
        for l in range(L):
            n = int(rng.integers(0, 40))
            rows = np.sort(rng.choice(40, n, replace=False))
            if kind == 0:
                cols = rng.random(n) * 1151 + 1e-9
            elif kind == 1:
                cols = np.frombuffer(rng.bytes(8 * n), dtype=np.float64).copy()
                cols = np.abs(np.where(np.isfinite(cols), cols, 1.0)) + 5e-324
            else:
                cols = np.array([special[int(i)] for i in rng.integers(0, len(special), n)])
            v[l, rows, 1] = cols
            first = np.arange(3, 40 * 8, 8) if kind == 0 else -np.frombuffer(rng.bytes(8 * 40), dtype=np.float64)
            v[l, :, 0] = np.where(np.isnan(first), -2.5, first)
            v[l, rows, 2] = rng.integers(0, 3, n) if kind == 0 else np.float32(rng.random(n)).astype(np.float64)
        for sem in (True, False):
            ref = json.dumps(io_utils.lane_records(v, sem), indent=4)
            assert io_utils.lane_json_text(v, sem) == ref, (trial, sem)
            io_utils.save_lane_seq_2d(v, str(tmp_path / 't.json'), sem)
            assert (tmp_path / 't.json').read_text() == ref
    assert io_utils.lane_json_text(np.zeros((72, 144, 3))) == '[]'
    from lanemapping_amd._lib import LanemapHipError
    with pytest.raises(LanemapHipError, match='cannot open'):
        io_utils.save_lane_seq_2d(np.zeros((1, 4, 3)), str(tmp_path / 'no_such_dir' / 't.json'))


def test_native_seqs_json_is_json_dump_byte_for_byte(tmp_path):
    """save_seqs_json (3-D polylines after the back-projection): native writer == json.dump(.., indent=4, cls=NpEncoder); other
    structures fall through to json.dump."""
    import json
    from lanemapping_amd import io_utils
    rng = np.random.default_rng(3)
    p = str(tmp_path / 'a.json')
    for trial in range(60):
        lines = []
        for l in range(int(rng.integers(0, 6))):
            n = int(rng.integers(1, 30))
            sq = rng.random((n, 3)) * 10.0 ** rng.integers(-6, 7, (n, 3)) * rng.choice([-1, 1], (n, 3))
            if trial % 3 == 0:
                sq = np.frombuffer(rng.bytes(8 * n * 3), dtype=np.float64).reshape(n, 3).copy()
                sq = np.where(np.isnan(sq), 1.5, sq)
            lines.append({'seq': sq, 'seq_len': n if l % 2 else np.int64(n), 'init_vertex': sq[0, :], 'end_vertex': sq[n - 1, :]})
        io_utils.save_seqs_json(lines, p)
        assert open(p).read() == json.dumps(lines, indent=4, cls=io_utils.NpEncoder), trial
    for other in ([{'a': 1}], [{'seq': np.ones((2, 3)), 'seq_len': 2, 'init_vertex': np.zeros(3), 'end_vertex': np.ones(3)}],
                  [{'seq_len': 1, 'seq': np.ones((1, 2)), 'init_vertex': np.ones(2), 'end_vertex': np.ones(2)}]):
        io_utils.save_seqs_json(other, p)
        assert open(p).read() == json.dumps(other, indent=4, cls=io_utils.NpEncoder)


@pytest.mark.parametrize('impl', ['product', 'oracle'])
def test_merge_lines_golden_g13(golden, tmp_path, impl):
    """merge_lines / downsample_seqs - the host C++ merger of the C-ABI (lm_merge_*) and the numpy oracle - vs the reference's own
    output on a 5-tile road (same-heading weave, reversed merge, new lines, retirement incl. the pop-while-enumerating skip):
    identical arrays."""
    import cases
    if impl == 'product':
        from lanemapping_amd import merge_lines as ml
    else:
        from oracle import merge_ref as ml
    g = golden('g13_merge.npz')
    merged = ml.merge_lines(cases.merge_case_files(str(tmp_path)))
    assert len(merged) == int(g['n'])
    for i, m in enumerate(merged):
        assert np.array_equal(m, g[f'merged_{i}'])
        assert np.array_equal(ml.downsample_seqs(m), g[f'down_{i}'])
    from lanemapping_amd import io_utils
    io_utils.save_seqs_list(merged, str(tmp_path / 'merged.txt'))
    io_utils.save_seqs_list([ml.downsample_seqs(m) for m in merged], str(tmp_path / 'merged.json'))
    rows = open(tmp_path / 'merged.txt').read().strip().split('\n')
    assert len(rows) == sum(len(m) for m in merged) and rows[0].endswith(' 0')
    back, lens, _, _ = io_utils.load_lane_seq(str(tmp_path / 'merged.json'), dim_coor=3)
    assert len(lens) == len(merged) and np.array_equal(back[0, :lens[0]], ml.downsample_seqs(merged[0]))


# ----------------------------------------------------------------------------------------------- f3: metrics
def test_metrics_golden_g14(golden):
    """Vertex and endpoint precision / recall / F1 vs the reference's metric_utils on seeded pred / GT pairs: identical."""
    import cases
    from lanemapping_amd import metric_utils as mu
    g = golden('g14_metrics.npz')
    for i, seed in enumerate(g['seeds']):
        label, pred, egt, epr = cases.metric_case(int(seed))
        assert np.array_equal(np.array(mu.cal_coor_measures(label, pred, 'conf', offset_thre=8 if i % 2 else 16)), g[f'coor_{i}'])
        assert np.array_equal(np.array(mu.eval_metric_endp_detector(epr, egt, r_thre=10)), g[f'endp_{i}'])
    z = np.zeros((1152, 1152), np.float32)
    assert mu.eval_metric_endp_detector(z, z) == (0., 0., 0, 0, 0, 0, 0)
    with pytest.raises(NotImplementedError):
        mu.cal_coor_measures(np.zeros((2, 144)), np.zeros((2, 144)), 'cls')


def test_entry_scripts_have_no_undefined_names():
    """bench.py / __graft_entry__.py are only executed end to end on the GPU box: catch NameErrors (a function using a name
    that exists only in another function's scope) here, on the CPU."""
    import ast
    import builtins
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for script in ('bench.py', '__graft_entry__.py'):
        tree = ast.parse(open(os.path.join(root, script)).read())
        glob = {n.id for n in ast.walk(tree) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store) and False}
        for n in tree.body:
            if isinstance(n, (ast.Import, ast.ImportFrom)):
                glob |= {(a.asname or a.name).split('.')[0] for a in n.names}
            elif isinstance(n, (ast.FunctionDef, ast.ClassDef)):
                glob.add(n.name)
            elif isinstance(n, ast.Assign):
                glob |= {t.id for t in n.targets if isinstance(t, ast.Name)}
        for fn in [n for n in tree.body if isinstance(n, ast.FunctionDef)]:
            local = set()
            for n in ast.walk(fn):
                if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Store):
                    local.add(n.id)
                elif isinstance(n, (ast.Import, ast.ImportFrom)):
                    local |= {(a.asname or a.name).split('.')[0] for a in n.names}
                elif isinstance(n, (ast.FunctionDef, ast.Lambda)):
                    local |= {a.arg for a in n.args.args + n.args.kwonlyargs}
                    if isinstance(n, ast.FunctionDef):
                        local.add(n.name)
                elif isinstance(n, ast.ExceptHandler) and n.name:
                    local.add(n.name)
                elif isinstance(n, ast.comprehension):
                    local |= {t.id for t in ast.walk(n.target) if isinstance(t, ast.Name)}
            used = {n.id for n in ast.walk(fn) if isinstance(n, ast.Name) and isinstance(n.ctx, ast.Load)}
            missing = sorted(u for u in used if u not in local and u not in glob and not hasattr(builtins, u))
            assert not missing, f'{script}:{fn.name} uses undefined names {missing}'


def test_bench_launcher_refuses_silent_single_gpu():
    """`bench.py --gpus N` never silently benchmarks one GPU (round-1 defect): a WORLD_SIZE that disagrees with --gpus, or fewer
    visible GPUs than ranks, stops the run with a message before any GPU call (so this runs on a CPU-only box)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'LANEMAP_BENCH_DEVICE')}
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2'], capture_output=True, text=True, timeout=300,
                       env=dict(env, WORLD_SIZE='1'), cwd=root)
    assert r.returncode != 0 and 'WORLD_SIZE=1' in r.stderr and not r.stdout.strip()
    if not torch.cuda.is_available() or torch.cuda.device_count() < 2:
        r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2'], capture_output=True, text=True, timeout=300,
                           env=env, cwd=root)
        assert r.returncode != 0 and 'visible' in r.stderr and not r.stdout.strip()


def test_rowref_lines_from_columns_matches_claim_loop():
    """RowSharNotReducRef.lines_from_columns (vectorised "a pixel claimed by a lower lane index wins") == the reference's
    set-based loop (row_shared_not_reduc_ref.py:487-516) followed by the same C++ tracing."""
    from lanemapping_amd import hostpost
    from lanemapping_amd.rowref import RowSharNotReducRef
    rng = np.random.default_rng(5)
    for _ in range(20):
        col = rng.integers(-1, 9, size=(12, 144)).astype(np.int32)
        col[rng.random((12, 144)) < 0.5] = -1
        lines = np.zeros((12, 144)) - 1.0
        taken = set()
        for c in range(12):
            for h in np.nonzero(col[c] >= 0)[0]:
                key = (int(h), int(col[c, h]))
                if key not in taken:
                    taken.add(key)
                    lines[c, h] = col[c, h] / 144 * 1152. + 4
        assert np.array_equal(RowSharNotReducRef.lines_from_columns(col, 144), hostpost.trace_lines(lines))


@pytest.mark.parametrize('seed', range(6))
def test_merge_lines_cpp_vs_oracle_random_roads(seed):
    """The streaming C++ merger vs the numpy oracle on seeded random roads: several parallel lanes cut into overlapping tiles,
    some pieces reversed, lanes that end or start mid-way, jittered vertices - identical merged and down-sampled arrays."""
    from lanemapping_amd import merge_lines as ml
    from oracle import merge_ref
    rng = np.random.RandomState(100 + seed)
    n_lanes, n_tiles = rng.randint(2, 6), rng.randint(3, 8)
    heading = rng.rand() * 2 * np.pi
    u, v = np.array([np.cos(heading), np.sin(heading), 0.0]), np.array([-np.sin(heading), np.cos(heading), 0.0])
    tiles = []
    for t in range(n_tiles):
        lines = []
        for l in range(n_lanes):
            if rng.rand() < 0.15:
                continue                                     # the lane is missing in this tile
            s0 = t * 40.0 + rng.rand() * 3
            n = rng.randint(12, 40)
            s = s0 + np.sort(rng.rand(n)) * 55.0             # 15 m overlap with the next tile
            pts = 1000.0 + s[:, None] * u + (l * 3.5 + 0.03 * rng.randn(n))[:, None] * v + np.array([0, 0, 1.0]) * (5 + 0.01 * s[:, None])
            if rng.rand() < 0.2:
                pts = pts[::-1]
            lines.append(pts)
        tiles.append(lines if len(lines) >= 2 else [])      # load_lane_seq drops single-line files
    m = ml.LineMerger()
    for lines in tiles:
        m.add_tile(lines)
    got = m.finish()

    class FakeLoad:                                          # the oracle reads files: feed it the same tiles
        def __init__(self):
            self.i = 0

        def __call__(self, name):
            lines = tiles[int(name)]
            return lines, [len(x) for x in lines], [x[0] for x in lines], [x[-1] for x in lines]
    orig = merge_ref._load
    merge_ref._load = FakeLoad()
    try:
        want = merge_ref.merge_lines([f'{i:04d}' for i in range(n_tiles)])
    finally:
        merge_ref._load = orig
    assert len(got) == len(want) and len(got) >= 1
    for a, b in zip(got, want):
        assert np.array_equal(a, b)
        assert np.array_equal(ml.downsample_seqs(a), merge_ref.downsample_seqs(b))


def test_torch_ops_registered_with_schema_and_fake_kernels():
    """Every hot-path op is a dispatcher-visible torch custom op (torch.ops.lanemap_hip.*) with a schema and a fake kernel that
    infers output shapes / strides without touching a device; device kernels are registered for cuda only - CPU tensors are refused
    by the dispatcher, there is no fallback implementation."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    from lanemapping_amd import torch_ops
    for name in torch_ops.OP_NAMES:
        op = getattr(torch.ops.lanemap_hip, name)
        assert op.default._schema is not None
    assert 'Tensor(a1!) fea_up_out' in str(torch.ops.lanemap_hip.fpn_encoder.default._schema)
    with FakeTensorMode():
        x = torch.empty((2, 16, 16, 64), device='cuda').permute(0, 3, 1, 2)
        y = torch.ops.lanemap_hip.conv2d_mfma(x, torch.empty((9, 128, 64), device='cuda'), 96, 3, 3, 2, 1, 1, None, None, None, 1)
        assert tuple(y.shape) == (2, 96, 8, 8) and y.stride(1) == 1
        y = torch.ops.lanemap_hip.conv3x3_winograd44(x, torch.empty((36, 8, 4, 64, 4), device='cuda'), 128, 2, None, None, None, 0)
        assert tuple(y.shape) == (2, 128, 16, 16)
        s = torch.ops.lanemap_hip.stem_conv7x7(torch.empty((3, 64, 48, 3), device='cuda', dtype=torch.uint8), torch.empty((7, 7, 3, 64), device='cuda'),
                                               torch.empty(64, device='cuda'), torch.empty(64, device='cuda'))
        assert tuple(s.shape) == (3, 64, 32, 24)
        t = torch.ops.lanemap_hip.bev_raster(torch.empty((100, 4), device='cuda'), [0, 40, 100], torch.zeros((2, 15)), 96, 80)
        assert tuple(t.shape) == (2, 96, 80, 3) and t.dtype == torch.uint8
        pc, ve, cc, ci, co = torch.ops.lanemap_hip.decode_proposals(torch.empty((1, 72, 2), device='cuda'), torch.empty((1, 72, 144, 3), device='cuda'),
                                                                    torch.empty((1, 72, 144, 10), device='cuda'), torch.empty((1, 72, 144, 10), device='cuda'),
                                                                    0.2, 2, 4)
        assert co.dtype == torch.float64 and ci.dtype == torch.int32 and tuple(cc.shape) == (1, 72, 144, 10)
        a = torch.ops.lanemap_hip.attention(torch.empty((648, 3072), device='cuda'), 2, 324, 16, 64, 0.125)
        assert tuple(a.shape) == (648, 1024)
    with pytest.raises((NotImplementedError, RuntimeError)):          # no CPU kernel: the dispatcher refuses
        torch.ops.lanemap_hip.tile_ingest(torch.zeros((1, 8, 8, 3), dtype=torch.uint8))
    # the two host ops run here: host C++ of the same library
    lanes, kept = torch.ops.lanemap_hip.polyline_assemble(torch.zeros((72, 2)), torch.zeros((72, 144)), torch.zeros((72, 144), dtype=torch.float64),
                                                          torch.zeros((144, 1152)), torch.zeros((0, 2), dtype=torch.int32), 0.3)
    assert tuple(lanes.shape) == (72, 144, 2) and lanes.dtype == torch.float64 and kept.shape[0] == 0
    # torch.library.opcheck (schema vs behaviour, fake kernel vs real outputs, dispatcher registrations) on the ops that can run here
    args = (torch.zeros((72, 2)), torch.zeros((72, 144)), torch.zeros((72, 144), dtype=torch.float64), torch.zeros((144, 1152)),
            torch.zeros((0, 2), dtype=torch.int32), 0.3)
    torch.library.opcheck(torch.ops.lanemap_hip.polyline_assemble.default, args, test_utils=('test_schema', 'test_faketensor'))
    idx = (torch.arange(512, dtype=torch.int32) * 2503) % (1136 * 1136)          # 512 valid flat indices of the cropped map
    torch.library.opcheck(torch.ops.lanemap_hip.endp_cluster.default, (idx, 1136, 8, 60, 500), test_utils=('test_schema', 'test_faketensor'))


def test_stage_ops_take_weights_and_a_stage_name():
    """The stage ops (what the modules' forward go through) take the stage's weights as a Tensor[] operand and its structure as a
    registered name - not id(module): the name comes from a process-wide counter (never recycled), the weights are visible to a
    tracer, and the fake kernels infer the output shapes from them without a device."""
    from torch._subclasses.fake_tensor import FakeTensorMode
    from lanemapping_amd import torch_ops
    from lanemapping_amd.boundary import build_net_from_config
    for name in ('fpn_encoder', 'vit_backbone', 'colprop_head'):
        sch = str(getattr(torch.ops.lanemap_hip, name).default._schema)
        assert 'Tensor[] weights' in sch and 'str stage' in sch and 'int module' not in sch, sch
    net = build_net_from_config('Proj_polyline_fpn_vit_vertex_2', device='cpu')
    enc = net.pcencoder.fpn           # (the stage module: FPNEncoder inside the PostProjector2 wrapper)
    n1, n2 = torch_ops.stage_name(enc), torch_ops.stage_name(net.backbone)
    assert n1 != n2 and n1 == torch_ops.stage_name(enc) and n1.split('#')[0] == type(enc).__name__
    w = torch_ops.stage_weights(enc)
    sd = enc.state_dict(keep_vars=True)
    assert len(w) == len(sd) and all(a is b for a, b in zip(w, sd.values()))          # every parameter and buffer, state_dict order
    with FakeTensorMode(allow_non_fake_inputs=True):
        x = torch.empty((2, 1152, 1152, 3), device='cuda', dtype=torch.uint8)
        up = torch.empty((2, 288, 288, 8), device='cuda').permute(0, 3, 1, 2)
        fea, bi, en = torch.ops.lanemap_hip.fpn_encoder(x, up, w, n1)
        assert tuple(fea.shape) == (2, 64, 144, 144) and tuple(bi.shape) == (2, 3, 1152, 1152) and tuple(en.shape) == (2, 1, 1152, 1152)
        y = torch.ops.lanemap_hip.vit_backbone(fea, torch_ops.stage_weights(net.backbone), n2)
        assert tuple(y.shape) == (2, 8, 144, 144)
    # a dead module's name is refused, never silently re-bound (id() could alias a recycled object)
    del net, enc
    import gc
    gc.collect()
    with pytest.raises(RuntimeError):
        with FakeTensorMode(allow_non_fake_inputs=True):
            torch.ops.lanemap_hip.fpn_encoder(torch.empty((1, 1152, 1152, 3), device='cuda', dtype=torch.uint8),
                                              torch.empty((1, 8, 288, 288), device='cuda'), w, n1)



def test_winograd44_weight_packers_layout():
    """Host-side packers of wino44_kernel (ops.pack_wino44*): U = G g G^T of F(4x4,3x3) against a direct fp64 evaluation (formed in fp64,
    rounded to fp32 once), CoutP padded to the 64-channel N tile with zero rows, and the per-wave-fragment order documented in
    include/lanemap_hip.h: wu_frag[xi][u][nt][lane][e] = U[xi][nt*32 + (lane & 31)][8 u + 4 (lane >> 5) + e]."""
    from lanemapping_amd import ops
    g = torch.Generator().manual_seed(11)
    cout, cin = 72, 48
    w = torch.randn((cout, cin, 3, 3), generator=g)
    wu = ops.pack_wino44(w)
    assert wu.shape == (36, 128, cin) and wu.dtype == torch.float32 and float(wu[:, cout:].abs().max()) == 0.0
    G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6], [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]])
    want = np.einsum('ij,ocjk,lk->iloc', G, w.double().numpy(), G).reshape(36, cout, cin)
    np.testing.assert_allclose(wu[:, :cout].numpy(), want, rtol=6e-8, atol=1e-12)       # formed in fp64, ONE rounding to fp32 (half an ulp)
    wf = ops.pack_wino44_fragments(wu)
    assert wf.shape == (36, cin // 8, 4, 64, 4)
    rng = np.random.default_rng(3)
    for _ in range(300):
        xi, u, nt, lane, e = (int(rng.integers(0, n)) for n in (36, cin // 8, 4, 64, 4))
        assert float(wf[xi, u, nt, lane, e]) == float(wu[xi, nt * 32 + (lane & 31), 8 * u + 4 * (lane >> 5) + e])
    # F(4x4,3x3) identity on one tile in fp64: A^T [(G g G^T) o (B^T d B)] A == the direct 3x3 correlation
    Bt = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0], [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], dtype=np.float64)
    At = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], dtype=np.float64)
    d, k = rng.standard_normal((6, 6)), rng.standard_normal((3, 3))
    y = At @ ((G @ k @ G.T) * (Bt @ d @ Bt.T)) @ At.T
    ref = np.array([[(d[i:i + 3, j:j + 3] * k).sum() for j in range(4)] for i in range(4)])
    np.testing.assert_allclose(y, ref, rtol=0, atol=1e-12)


def test_bench_host_budget_per_rank():
    """bench.py's per-rank host budget (pure function, decided before the GPU is touched): an 8-rank launch on a box with 16 usable cores
    gives every rank its own 2 cores, a 1-thread pool and HIP graphs; 2 ranks get 8 each without graphs; a roomy node is left alone;
    `--host-cores` wins over the automatic choice; one rank never pins itself."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location('bench_mod', os.path.join(root, 'bench.py'))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    allowed = list(range(256))
    seen = set()
    for r in range(8):
        hb = bench.host_budget(8, 8, r, 16, allowed, None, False, 'fused')
        assert hb['auto'] and hb['k'] == 2 and hb['graphs'] and len(hb['cores']) == 2 and not (seen & set(hb['cores']))
        seen |= set(hb['cores'])
    hb = bench.host_budget(2, 2, 1, 16, allowed, None, False, 'fused')
    assert hb['auto'] and hb['k'] == 8 and not hb['graphs'] and hb['cores'] == list(range(8, 16))
    assert bench.host_budget(8, 8, 3, 128, allowed, None, False, 'fused') == {'k': None, 'cores': None, 'graphs': False, 'auto': False, 'numa_node': None}
    # NUMA-aware pick: 8 GPUs on two nodes (0-3 on node 0 = cpus 0..63 and 128..191, 4-7 on node 1), 16 usable cores per node
    topo = {'gpu_node': [0, 0, 0, 0, 1, 1, 1, 1], 'node_cpus': {0: list(range(0, 64)) + list(range(128, 192)), 1: list(range(64, 128)) + list(range(192, 256))}}
    mask = list(range(0, 16)) + list(range(64, 80))
    got = [bench.host_budget(8, 8, r, 32, mask, None, False, 'fused', topo=topo) for r in range(8)]
    assert [g['numa_node'] for g in got] == [0, 0, 0, 0, 1, 1, 1, 1] and all(g['k'] == 4 for g in got)
    assert got[0]['cores'] == [0, 1, 2, 3] and got[3]['cores'] == [12, 13, 14, 15] and got[4]['cores'] == [64, 65, 66, 67] and got[7]['cores'] == [76, 77, 78, 79]
    flat = [c for g in got for c in g['cores']]
    assert len(set(flat)) == 32                                                                  # disjoint over the ranks
    # a node without enough usable cores for its ranks: fall back to the contiguous slice of the mask
    hb = bench.host_budget(8, 8, 5, 32, list(range(0, 32)), None, False, 'fused', topo=topo)
    assert hb['numa_node'] is None and hb['cores'] == [20, 21, 22, 23]
    assert bench.gpu_numa_topology() is None or isinstance(bench.gpu_numa_topology()['gpu_node'], list)   # (no amdgpu card in the build container)
    hb = bench.host_budget(8, 8, 3, 128, allowed, 4, False, 'tiles')
    assert not hb['auto'] and hb['k'] == 4 and hb['graphs'] and hb['cores'] == [12, 13, 14, 15]
    assert not bench.host_budget(8, 8, 0, 16, allowed, None, True, 'fused')['graphs']            # --no-graphs
    assert not bench.host_budget(8, 8, 0, 16, allowed, None, False, 'lidar')['graphs']           # the LiDAR path is not captured
    assert bench.host_budget(1, 1, 0, 16, allowed, None, False, 'fused')['cores'] is None
    assert bench.host_budget(1, 1, 0, 16, [0, 1, 2], 8, False, 'fused')['k'] == 3                # never more than the mask allows


# ----------------------------------------------------------------------------------------------- f4: the PNG reader's DEFLATE decoder
def _zlib_cases(rng):
    """(name, raw bytes) - what DEFLATE has to cope with: incompressible noise, runs, short alphabets (two literals per table entry),
    long repeats at every small distance, text-like data with long codes, empty input."""
    n = 70000
    yield 'noise', rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    yield 'zeros', bytes(n)
    yield 'short alphabet', (rng.integers(0, 5, n, dtype=np.uint8) * 50).tobytes()
    yield 'skewed', np.minimum(rng.geometric(0.3, n), 255).astype(np.uint8).tobytes()          # code lengths 1 ... 15: sub-tables
    for d in (1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 31, 255, 32768):
        pat = rng.integers(0, 256, d, dtype=np.uint8).tobytes()
        yield f'period {d}', (pat * (n // d + 2))[:n] if d < 32768 else pat + rng.integers(0, 256, 70, dtype=np.uint8).tobytes() + pat
    yield 'tile-like', (np.abs(rng.normal(0, 9, (n // 3, 3))).astype(np.uint8) * (rng.random((n // 3, 1)) < 0.3)).tobytes()
    yield 'one byte', b'\x07'
    yield 'empty', b''


def test_zlib_inflate_matches_zlib_on_every_block_type():
    """lm_zlib_inflate (csrc/inflate.h) == zlib.decompress on stored / fixed / dynamic blocks, every level and strategy, multi-block
    streams (flush points) and exact-size / oversize output buffers."""
    import zlib
    from lanemapping_amd import png_io
    from lanemapping_amd._lib import LanemapHipError
    rng = np.random.default_rng(21)
    seen = 0
    for name, raw in _zlib_cases(rng):
        for level in (0, 1, 4, 6, 9):
            for strat in (zlib.Z_DEFAULT_STRATEGY, zlib.Z_FIXED, zlib.Z_HUFFMAN_ONLY, zlib.Z_RLE, zlib.Z_FILTERED):
                for mem in (1, 9):
                    c = zlib.compressobj(level, zlib.DEFLATED, 15, mem, strat)
                    half = len(raw) // 2
                    z = c.compress(raw[:half]) + c.flush(zlib.Z_FULL_FLUSH) + c.compress(raw[half:]) + c.flush()
                    assert png_io.zlib_inflate(z, len(raw)) == raw, (name, level, strat, mem)
                    assert png_io.zlib_inflate(z, len(raw) + 1000) == raw, (name, level, strat, mem)
                    seen += 1
                    if raw:
                        with pytest.raises(LanemapHipError, match='larger'):
                            png_io.zlib_inflate(z, len(raw) - 1)
    assert seen == 22 * 50
    for wbits in (9, 12):                                       # small windows (CINFO < 7)
        c = zlib.compressobj(6, zlib.DEFLATED, wbits)
        raw = bytes(rng.integers(0, 3, 5000, dtype=np.uint8))
        assert png_io.zlib_inflate(c.compress(raw) + c.flush(), 5000) == raw


def test_zlib_inflate_refuses_malformed_streams():
    """Hand-made bad streams are refused by name; every truncation of a good stream and random bit damage end in the same bytes zlib
    produces or in an error - never a crash, never different data accepted."""
    import zlib
    from lanemapping_amd import png_io
    from lanemapping_amd._lib import LanemapHipError
    rng = np.random.default_rng(22)

    def bits_to_stream(bits):
        """'0'/'1' string in stream order (first bit = LSB of the first byte) behind a zlib header; no Adler trailer."""
        bits += '0' * (-len(bits) % 8)
        return b'\x78\x9c' + bytes(int(bits[i:i + 8][::-1], 2) for i in range(0, len(bits), 8))

    def field(v, n):
        return ''.join(str((v >> i) & 1) for i in range(n))

    cases = {
        'bad header': b'\x79\x9c' + b'\0' * 8,                                     # CM = 9
        'preset dictionary': b'\x78\xbb' + b'\0' * 8,                              # FDICT (0x78bb % 31 == 0)
        'block type 3': bits_to_stream('1' + field(3, 2)) + b'\0\0\0\0',
        'stored block length': b'\x78\x9c\x01\x05\x00\x00\x00hello' + b'\0\0\0\0',
        # dynamic block whose code-length code is over-subscribed: HLIT 0, HDIST 0, HCLEN 15 (19 lengths), all of length 1
        'over-subscribed': bits_to_stream('1' + field(2, 2) + field(0, 5) + field(0, 5) + field(15, 4) + field(1, 3) * 19),
        # fixed block: length symbol 257 (7-bit code 0000001) + distance code 0 (5 bits) with nothing in the window
        'distance before the start': bits_to_stream('1' + field(1, 2) + '0000001' + '00000'),
        # fixed block: literal/length symbol 286 (8-bit code 11000110) never appears in a valid stream
        'bad literal / length': bits_to_stream('1' + field(1, 2) + '11000110'),
        # fixed block: one literal, a length, then distance symbol 30 (5-bit code 11110)
        'bad distance': bits_to_stream('1' + field(1, 2) + '00110000' + '0000001' + '11110'),
    }
    for word, z in cases.items():
        with pytest.raises(zlib.error):
            zlib.decompress(z)
        with pytest.raises(LanemapHipError, match=word):
            png_io.zlib_inflate(z, 1000)

    raw = (np.abs(rng.normal(0, 20, 6000)).astype(np.uint8)).tobytes() + b'abcabcabc' * 40 + bytes(700)
    fixed = zlib.compressobj(9, zlib.DEFLATED, 15, 9, zlib.Z_FIXED)
    for good in (zlib.compress(raw, 6), zlib.compress(raw, 0), fixed.compress(raw) + fixed.flush()):
        assert png_io.zlib_inflate(good, len(raw)) == raw
        for cut in list(range(0, 64)) + list(range(len(good) - 64, len(good))) + [int(c) for c in rng.integers(64, len(good) - 64, 200)]:
            with pytest.raises(LanemapHipError, match='truncated|corrupt'):
                png_io.zlib_inflate(good[:cut], len(raw))
        with pytest.raises(LanemapHipError, match='data after the end'):
            png_io.zlib_inflate(good + b'\0', len(raw))
        same = refused = 0
        for _ in range(1500):
            data = bytearray(good)
            for _ in range(int(rng.integers(1, 4))):
                data[int(rng.integers(0, len(data)))] ^= 1 << int(rng.integers(0, 8))
            try:
                want = zlib.decompress(bytes(data))
            except zlib.error:
                want = None
            try:
                got = png_io.zlib_inflate(bytes(data), len(raw) + 300)
            except LanemapHipError:
                got = None
            if want is not None and len(want) <= len(raw) + 300:
                assert got == want                      # (the Adler check makes this all but unreachable; equality when it happens)
                same += 1
            else:
                assert got is None
                refused += 1
        assert refused > 1400
