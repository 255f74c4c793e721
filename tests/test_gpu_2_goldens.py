"""GPU parity tests (-m gpu), file 2 of 4: the HIP path against the golden vectors generated from the reference (G2-G17) and against
the oracle on whole tiles.  Final coordinates / confidences within 1e-4 absolute, integer / index outputs bit-exact."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cases
from gpu_common import ROOT, _close, _lidar_module, _rowref_head
from lanemapping_amd import synth

pytestmark = pytest.mark.gpu


# ----------------------------------------------------------------------------------------------- goldens
def test_fpn_golden_g2(dev, net, golden):
    g = golden('g2_fpn.npz')
    x = torch.from_numpy(synth.bev_batch([int(s) for s in g['seeds']], int(g['size']))).to(dev)
    with torch.no_grad():
        out = net.pcencoder({'proj': x})
    for name, o in zip(('fea', 'fea_up', 'bi_seg', 'endp'), out):
        _close(o, g[name], 1e-4, name)


def test_vit_golden_g3(dev, net, golden):
    g = golden('g3_vit.npz')
    with torch.no_grad():
        y = net.backbone(torch.from_numpy(cases.vit_input(int(g['input_seed']))).to(dev))
    _close(y, g['out'], 1e-4, 'vit')


def test_head_golden_g4(dev, net, golden):
    g = golden('g4_head.npz')
    x, x_up = cases.head_inputs(int(g['input_seed']))
    with torch.no_grad():
        out = net.heads(torch.from_numpy(x).to(dev), torch.from_numpy(x_up).to(dev), None)
    for k in ('proposal_conf', 'ext2', 'cls2', 'offset2', 'orient'):
        _close(out[k], g[k], 1e-4, k)
    # class indices: bit-exact wherever the reference's own top-2 margin exceeds the fp32 noise floor
    idx = out['cls2'].argmax(-1).cpu().numpy()
    ref_idx = g['cls2'].argmax(-1)
    safe = g['cls2_margin'] > 1e-3
    assert np.array_equal(idx[safe], ref_idx[safe])
    assert (idx != ref_idx).sum() == 0, f'{(idx != ref_idx).sum()} argmax flips inside the noise margin'


def test_decode_golden_g5(dev, net, golden):
    g = golden('g5_decode.npz')
    raw = cases.decode_inputs(int(g['input_seed']), batch=int(g['batch']))
    out = {k: torch.from_numpy(v).to(dev) for k, v in raw.items()}
    out['orient'] = out['orient'].contiguous(memory_format=torch.channels_last)
    d = net.heads.get_exist_coor_endp_dict(out)
    assert np.array_equal(d['prop_v_ext'].numpy().astype(np.uint8), g['prop_v_ext'])
    assert np.array_equal(d['orient'].numpy().astype(np.uint8), g['orient'])
    assert np.array_equal(d['semantic_seg'].numpy().astype(np.uint8), g['semantic_seg'])
    _close(d['prop_conf'], g['prop_conf'], 1e-5, 'prop_conf')
    _close(d['prop_cls_conf'], g['prop_cls_conf'], 1e-5, 'prop_cls_conf')
    np.testing.assert_allclose(d['cls_offset'].numpy(), g['cls_offset'], rtol=0, atol=1e-6)
    _close(d['bi_seg'][:, 3::8, :], g['bi_seg_rows'], 1e-5, 'bi_seg')
    for b in range(2):
        assert np.array_equal(np.stack(np.nonzero(d['endp'][b].numpy()), axis=1), g[f'endp{b}'])


def test_segmentor_golden_g7(dev, golden):
    from lanemapping_amd.decode import segmentor_decode
    g = golden('g7_segmentor.npz')
    raw = cases.decode_inputs(int(g['input_seed']), batch=1)
    r = segmentor_decode(torch.from_numpy(raw['semantic_seg']).to(dev), torch.from_numpy(raw['endp_est']).to(dev), 0.1)
    assert np.array_equal(r['seg'].numpy().astype(np.uint8), g['seg'])
    assert np.array_equal(np.stack(np.nonzero(r['endp'][0].numpy()), axis=1), g['endp'])


def test_end_to_end_golden_g10(dev, net, golden):
    """One full 1152^2 tile through Detector1stage (config 2) vs the reference's own end-to-end run."""
    g = golden('g10_e2e.npz')
    x = torch.from_numpy(synth.bev_batch([int(g['tile_seed'])], 1152)).to(dev)
    with torch.no_grad():
        raw = net.forward_raw({'proj': x})
        for k, gk in (('proposal_conf', 'proposal_conf'), ('ext2', 'ext2'), ('cls2', 'cls2'), ('offset2', 'offset2'),
                      ('orient', 'orient_logits')):
            _close(raw[k], g[gk], 1e-4, k)
        o = net({'proj': x})
    # Integer outputs.  The summation order of the HIP convolutions differs from the reference's MKL-DNN order, so a
    # class decision may legitimately flip only where the REFERENCE's own decision margin (stored in the golden:
    # distance to a tie or to the threshold, < 1e-4) is inside fp32 noise; everywhere else the match must be exact.
    def flips_inside_noise(mine, ref, low_idx, name, budget):
        bad = np.flatnonzero(mine.reshape(-1) != ref.reshape(-1))
        outside = np.setdiff1d(bad, low_idx)
        assert outside.size == 0, f'{name}: {outside.size} mismatches where the reference margin is >= 1e-4'
        assert bad.size <= budget, f'{name}: {bad.size} noise-margin flips (budget {budget})'
        print(f'{name}: {bad.size} flips, all among the {low_idx.size} reference low-margin entries of {ref.size}')
    flips_inside_noise(o['prop_v_ext'].numpy().astype(np.uint8)[0], g['prop_v_ext'][0], g['ext_lowmargin'], 'prop_v_ext', 0)
    flips_inside_noise(o['orient'].numpy().astype(np.uint8)[0], g['orient'][0], g['orient_lowmargin'], 'orient', 1)
    flips_inside_noise(o['semantic_seg'].numpy().astype(np.uint8)[0], g['semantic_seg'][0], g['sem_lowmargin'], 'semantic_seg', 32)
    cls_idx = net.heads._compact['cls_idx'].cpu().numpy()[0]
    assert np.array_equal(cls_idx, g['cls2'][0].argmax(-1)), 'column-bin argmax must match the reference exactly'
    # cls_offset = bin index + offset2[bin] + proposal origin: exact integers plus one fp32 regression output, so its error
    # IS offset2's; with the seeded random weights |offset2| reaches ~30 (a trained head keeps it inside one bin), hence the
    # same "1e-4 of the tensor scale" bound as for offset2 itself (= 1e-4 absolute for a head with |offset2| <= 1)
    off_scale = max(1.0, float(np.abs(g['offset2']).max()))
    np.testing.assert_allclose(o['cls_offset'].numpy(), g['cls_offset'], rtol=0, atol=1e-4 * off_scale)
    _close(o['prop_conf'], g['prop_conf'], 1e-4, 'prop_conf')
    assert np.array_equal(np.stack(np.nonzero(o['endp'][0].numpy()), axis=1), g['endp'])
    # Polyline assembly is a discontinuous function of its inputs (greedy tracing, int() truncation, confidence
    # comparisons): on this random-weight tile (48 spurious lines) a 1e-7 perturbation of the decode outputs already
    # changes the reference's own result (tests/test_boundary_cpu.py::test_postproc_is_chaotic_on_g10).  Vertex parity is
    # therefore pinned stage-wise: (a) the C++ assembly is bit-exact on the reference's decode outputs (CPU tests, G6 +
    # G10), (b) here: the product's polylines equal the oracle's assembly run on the product's own decode outputs.
    from oracle import postproc_ref
    c = net.heads._compact
    V = o['lane_maps']['cls_offset_smooth'][0]
    Vo, Eo, _ = postproc_ref.assemble_tile(c['prop_conf'][0, :, 1].cpu().numpy(), c['prop_v_ext'][0].cpu().numpy(),
                                           c['cls_offset'][0].cpu().numpy(), c['bi_seg'][0].cpu().numpy(), o['endp'][0].numpy())
    assert np.array_equal(V, Vo), 'product polylines must equal the oracle assembly on identical decode outputs'
    assert np.array_equal(o['lane_maps']['endp_by_cls'][0], Eo)
    ref_lines = {tuple(np.round(l[:, 0], 3)) for l in g['cls_offset_smooth'] if (l[:, 0] > 0).sum() >= 2}
    my_lines = {tuple(np.round(l[:, 0], 3)) for l in V if (l[:, 0] > 0).sum() >= 2}
    print(f'polylines identical to the reference run: {len(ref_lines & my_lines)} of {len(ref_lines)} (informational)')
    # accuracy-level view of the same thing, with the reference's own vertex metric (metric_utils.cal_coor_measures):
    # the reference's polylines as ground truth, 2 px buffer
    from lanemapping_amd import metric_utils
    acc, rec, f1, *_ = metric_utils.cal_coor_measures(np.where(g['cls_offset_smooth'][:, :, 0] > 0, g['cls_offset_smooth'][:, :, 0], -1.0),
                                                      np.where(V[:, :, 0] > 0, V[:, :, 0], -1.0), 'conf', offset_thre=2)
    print(f'vertex precision / recall / F1 vs the reference polylines at 2 px: {acc:.4f} / {rec:.4f} / {f1:.4f}')
    assert f1 > 0.9


def test_net_vs_oracle_batch2(dev, net, synth_sd):
    """Seeded inputs not covered by a golden: HIP raw outputs vs the oracle at batch 2, 576^2 tiles."""
    from oracle import net_ref
    x = torch.from_numpy(synth.bev_batch([101, 102], 576))
    with torch.no_grad():
        fea, fea_up, bi_seg, endp = net_ref.fpn_forward(synth_sd, x)
        mine = net.pcencoder({'proj': x.to(dev)})
    for name, a, b in zip(('fea', 'fea_up', 'bi_seg', 'endp'), mine, (fea, fea_up, bi_seg, endp)):
        _close(a, b, 1e-4, name)


def test_u8_tile_path_bit_identical(dev, net):
    """The pipeline hands BEV tiles over as u8 HWC (rasteriser / PNG reader output): lm_stem_conv7x7_bn_relu_u8 applies u8 / 255 while
    staging, so the whole net gives the SAME BITS as the reference's f32 planar tensor (load_img: to_tensor(u8))."""
    from lanemapping_amd import ops
    u8 = torch.from_numpy(np.stack([synth.bev_tile_u8(s, 1152) for s in (71, 72)])).to(dev)          # [2,1152,1152,3]
    f32 = ops.tile_ingest(u8)
    assert torch.equal(f32.cpu(), torch.from_numpy(synth.bev_batch([71, 72], 1152)))
    P = net.pcencoder.fpn.packed()
    assert torch.equal(ops.stem(u8, P['stem_w'], P['stem_s'], P['stem_b']), ops.stem(f32, P['stem_w'], P['stem_s'], P['stem_b']))
    with torch.no_grad():
        a = net.forward_raw({'proj': u8})
        b = net.forward_raw({'proj': f32})
    for k in b:
        assert torch.equal(a[k], b[k]), k
    # rasteriser: the u8-only output equals the u8 tile of the two-output call, whose f32 tile is u8 / 255
    pts = torch.from_numpy(synth.las_points(91, 300000)).to(dev)
    par = [ops.make_raster_params(local_min_ele=-0.5, ele_reso=0.02)]
    both = ops.bev_raster_batch(pts, [0, pts.shape[0]], par, want_u8=True)
    only = ops.bev_raster_batch(pts, [0, pts.shape[0]], par, u8_only=True)
    assert torch.equal(only, both[1]) and torch.equal(ops.tile_ingest(only), both[0])


def test_rowref_head_golden_g8(dev, golden):
    """Config 4 head: forward (incl. the shrinking-range scatter), decode and label-free line assembly vs the reference."""
    g = golden('g8_rowref.npz')
    head = _rowref_head(dev)
    x = torch.from_numpy(cases.head_inputs(int(g['input_seed']), batch=2)[0]).to(dev)
    with torch.no_grad():
        out = head(x)
        dec = head.get_exist_coor_endp_dict(out)
    assert head._last['selected'].all()
    for c in range(12):
        _close(out[f'ext_{c}'][:, :, 0].mean(dim=1), g[f'ext_mean_{c}'], 1e-5, f'ext_mean_{c}')
        _close(out[f'ext2_{c}'], g[f'ext2_{c}'], 1e-4, f'ext2_{c}')
        arg = out[f'cls2_{c}'].argmax(dim=2).cpu().numpy()
        safe = g[f'cls2_margin_{c}'] > 1e-4
        assert np.array_equal(arg[safe], g[f'cls2_arg_{c}'][safe]), f'cls2_{c} argmax'
        _close(out[f'cls2_{c}'].max(dim=2).values, g[f'cls2_max_{c}'], 1e-4, f'cls2_max_{c}')
    assert np.array_equal(dec['conf'].numpy().astype(np.uint8), g['conf'])
    assert np.array_equal(dec['cls'].numpy().astype(np.uint8), g['cls'])
    lines = head.predict_lines()
    for b in range(2):
        assert np.array_equal(lines[b], g['pred_lines'][b])
    # second pass: only part of the lanes passes the existence gate (thr_ext = 0.5)
    head.thr_ext = 0.5
    with torch.no_grad():
        out = head(x)
        dec = head.get_exist_coor_endp_dict(out)
    assert np.array_equal(head._last['selected'], g['t5_selected'])
    for c in range(12):
        _close(out[f'ext2_{c}'], g[f't5_ext2_{c}'], 1e-4, f't5_ext2_{c}')
    assert np.array_equal(dec['conf'].numpy().astype(np.uint8), g['t5_conf'])
    assert np.array_equal(dec['cls'].numpy().astype(np.uint8), g['t5_cls'])


def test_lidar_tail_golden_g11(dev, golden):
    import cases
    g = golden('g11_lidar_tail.npz')
    m, _ = _lidar_module(dev, cases.small_lidar_cfg(), int(g['weight_seed']))
    dense = torch.from_numpy(cases.lidar_tail_input(int(g['input_seed'])))
    with torch.no_grad():
        outs = m.dense_tail(torch.flip(dense, dims=[2]).to(dev))
    for name, o in zip(('fea', 'fea_up', 'bi_seg', 'endp'), outs):
        _close(o, g[name], 1e-4, name)


# ----------------------------------------------------------------------------------------------- edge cases
def test_edge_tiles_empty_and_saturated(dev, net, synth_sd):
    """An all-empty tile (no LiDAR return at all), a saturated one and batch 1: raw outputs vs the oracle, the full forward
    runs, and the polylines equal the oracle assembly on the product's own decode outputs."""
    from oracle import net_ref, postproc_ref
    x = torch.zeros((2, 3, 1152, 1152))
    x[1] = 1.0
    with torch.no_grad():
        ref = net_ref.detector_forward(synth_sd, x)
        raw = net.forward_raw({'proj': x.to(dev)})
        for k in ('proposal_conf', 'ext2', 'cls2', 'offset2', 'orient', 'semantic_seg', 'endp_est'):
            _close(raw[k], ref[k], 1e-4, k)
        for b in range(2):                                   # batch 1 == the same tile inside a batch of 2
            one = net.forward_raw({'proj': x[b:b + 1].to(dev)})
            for k in ('proposal_conf', 'cls2', 'semantic_seg'):
                assert torch.equal(one[k][0], raw[k][b]), f'{k}: batch-size dependent result'
        o = net({'proj': x.to(dev)})
    c = net.heads._compact
    for b in range(2):
        Vo, _, _ = postproc_ref.assemble_tile(c['prop_conf'][b, :, 1].cpu().numpy(), c['prop_v_ext'][b].cpu().numpy(),
                                              c['cls_offset'][b].cpu().numpy(), c['bi_seg'][b].cpu().numpy(), o['endp'][b].numpy())
        assert np.array_equal(o['lane_maps']['cls_offset_smooth'][b], Vo)


def test_multi_stream_pipeline_bitwise_equals_single_stream(dev, net):
    """The product path bench.py times splits a batch over 4 HIP streams and 4 TilePipelines that share one net (packed weights,
    per-stream workspaces): its lanes and endpoints must equal a single-stream run on the same tiles BITWISE, step after step."""
    from lanemapping_amd.pipeline import TilePipeline
    B, ns = 8, 4
    tiles = torch.from_numpy(synth.bev_batch([4100 + i for i in range(B)], 1152)).to(dev)
    pipes = [TilePipeline(net) for _ in range(ns)]
    streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(ns - 1)]
    ref = TilePipeline(net).run_batch(tiles)
    torch.cuda.synchronize()
    for rep in range(3):
        futs = []
        for si in range(ns):
            with torch.cuda.stream(streams[si]):
                pipes[si].submit(tiles[2 * si:2 * si + 2])
        for si in range(ns):
            with torch.cuda.stream(streams[si]):
                futs += pipes[si].flush()
        got = [f.result() for f in futs]
        assert len(got) == B
        for t, ((la, ea), (lb, eb)) in enumerate(zip(got, ref)):
            assert np.array_equal(la, lb) and np.array_equal(ea, eb), f'tile {t}, repetition {rep}'


# How many stability-screened tiles may differ from the reference in a SOFT endpoint decision (one the reference itself changes under a
# 1e-4 perturbation: `margin_ok` below) before the test fails.  Not a silent budget: every run prints the count, and with
# LANEMAP_PARITY_LOG=<file> appends it to that file (profiles/r5_g15_g17_exact_counts.txt holds the counts of the default route and of
# LANEMAP_WINO_F44=0 on an MI355X: 9 of 10 / 4 of 4 on both routes, so the budget is ONE tile).
_SOFT_BUDGET = 1


def _log_exact(tag, exact, seeds):
    route = ('direct (LANEMAP_WINO_F44=0)' if os.environ.get('LANEMAP_WINO_F44', '1') == '0' else
             'F(4x4) fp16x2 split second line (LANEMAP_WINO_SPLIT=1)' if os.environ.get('LANEMAP_WINO_SPLIT', '0') != '0' else 'F(4x4) default')
    line = (f'{tag} [{route}]: {sum(exact)} of {len(seeds)} stability-screened tiles identical to the reference in EVERY endpoint; soft-decision '
            f'differences on seeds {[s for s, e in zip(seeds, exact) if not e]}')
    print(line)
    if os.environ.get('LANEMAP_PARITY_LOG'):
        with open(os.environ['LANEMAP_PARITY_LOG'], 'a') as f:
            f.write(line + '\n')


def _g15_net(dev, synth_sd, g):
    from lanemapping_amd.boundary import build_net_from_config
    net = build_net_from_config('Proj_polyline_fpn_vit_vertex_2', device='cpu')
    sd = {k: v.clone() for k, v in synth_sd.items()}
    for k, gain in zip(g['gain_keys'], g['gain_values']):
        sd[str(k)] = sd[str(k)] * float(gain)
    net.load_state_dict(sd, strict=True)
    return net.to(dev), sd


def _check_stable_golden(g, i, b, c, o, res, name):
    """Tile i of a stability-screened golden (G15 / G17) against entry b of a batch result: decode outputs within ABSOLUTE 1e-4,
    existence classes exactly, the final polylines - which vertices exist, their semantics - exactly (columns within 8e-4 px), both
    from the net's own post-processing and from the TilePipeline.  Endpoints are MARGIN-AWARE like the class flips of G10: the pick
    (top-K scores, clustering, sample nearest the centroid) is a near-tie wherever a pixel enters or leaves the top K, so the golden
    records which endpoints the reference keeps under its own 1e-4 perturbations (`*_firm`) and everything any of those runs produced
    (`*_any`): firm endpoints must be there, nothing outside the union may appear.  Returns True when the tile is exact in everything."""
    d_off = np.abs(c['cls_offset'][b].cpu().numpy() - g[f'cls_offset{i}'])
    # column bins are MARGIN-AWARE like G10's class flips (round 6): cls_offset = bin + offset may differ by whole bins only where the
    # REFERENCE's own top-two cls2 logits are closer than the tolerance (`cls_margin`, stored by make_golden.py), at most twice per tile;
    # everywhere else the bound is absolute 1e-4.  (G17 cloud 3103 has one such cell - proposal 63, row 107, margin 2.1e-5, existence
    # class 2, in no polyline - which the direct-convolution route decides the other way: profiles/r6_g17_direct_route.txt)
    soft = g[f'cls_margin{i}'] < 1e-4
    flips = np.argwhere((d_off > 1e-4) & soft)
    err_off = float(d_off[~((d_off > 1e-4) & soft)].max())
    err_conf = float(np.abs(c['prop_conf'][b].cpu().numpy() - g[f'prop_conf{i}']).max())
    assert err_off <= 1e-4 and err_conf <= 1e-4 and len(flips) <= 2, (name, err_off, err_conf, flips.tolist())
    if len(flips):
        print(f'{name}: column bin differs in {len(flips)} cell(s) where the reference margin is < 1e-4: {flips.tolist()}')
    assert np.array_equal(c['prop_v_ext'][b].cpu().numpy().astype(np.uint8), g[f'prop_v_ext{i}']), name
    if f'prop_cls_conf{i}' in g.files:
        # absolute 1e-4 end to end on the class confidences (softmax of cls2; cells with a soft bin decision included: the two top
        # confidences are then equal within the margin) and on the foreground probability of the rows the assembly reads (8 h + 3, h even)
        err_cc = float(np.abs(c['prop_cls_conf'][b].float().cpu().numpy() - g[f'prop_cls_conf{i}']).max())
        rows = c['bi_seg_rows'][b].float().cpu().numpy()
        err_bs = float(np.abs(rows[0::2] - g[f'bi_seg_rows{i}']).max())
        assert err_cc <= 1e-4 and err_bs <= 1e-4, (name, 'prop_cls_conf', err_cc, 'bi_seg rows', err_bs)

    def rows(a):
        return {tuple(int(v) for v in r) for r in np.asarray(a).reshape(-1, 2)}

    def margin_ok(mine, base, firm, union, what):
        mine, base, firm, union = rows(mine), rows(base), rows(firm), rows(union)
        assert firm <= mine <= union, f'{name}: {what}: firm endpoints missing {sorted(firm - mine)}, outside the reference\'s 1e-4 runs {sorted(mine - union)}'
        assert len(mine ^ base) <= 2 * (len(base) - len(firm)), f'{name}: {what} differ from the reference in more places than it has soft decisions'
        return mine == base
    exact = margin_ok(np.stack(np.nonzero(o['endp'][b].numpy()), axis=1), g[f'endp{i}'], g[f'endp_firm{i}'], g[f'endp_any{i}'], 'decode endpoints')
    W = g[f'V{i}']
    for V, E in ((o['lane_maps']['cls_offset_smooth'][b], np.stack(np.nonzero(o['lane_maps']['endp_by_cls'][b]), axis=1)), res[b]):
        assert np.array_equal(V[:, :, 0] > 0, W[:, :, 0] > 0), f'{name}: vertex set differs from the reference'
        assert np.array_equal(V[:, :, 1], W[:, :, 1]), f'{name}: semantics differ from the reference'
        assert float(np.abs(V[:, :, 0] - W[:, :, 0]).max()) <= 8e-4, name
        exact = margin_ok(E, g[f'E{i}'], g[f'E_firm{i}'], g[f'E_any{i}'], 'kept endpoints') and exact
    print(f'{name}: cls_offset err {err_off:.2e}, prop_conf err {err_conf:.2e}, '
          f'{int((np.count_nonzero(W[:, :, 0] > 0, axis=1) >= 2).sum())} lines identical to the reference, endpoints '
          f'{"identical" if exact else "differ inside the reference margin"} ({len(rows(g[f"endp_firm{i}"]))} of {len(rows(g[f"endp{i}"]))} firm)')
    return exact


def test_end_to_end_stable_golden_g15(dev, golden, synth_sd):
    """Golden G15: tiles SCREENED so that the reference's own final polylines are invariant under a 1e-5 input perturbation, with
    an offset-regression layer that keeps vertex columns inside their bin (|offset2| <= 0.1).  The HIP path reproduces the reference
    end to end: cls_offset / prop_conf (and, on the first four tiles, prop_cls_conf / bi_seg rows) within ABSOLUTE 1e-4 (north_star's
    bound), existence classes exactly, endpoint pixels exactly on all but at most _SOFT_BUDGET tiles (margin-aware, see
    _check_stable_golden), and the final cls_offset_smooth - which vertices exist, their semantics, the kept endpoints - exactly, columns within 8e-4 px
    (= 1e-4 in column-bin units x 8 px).  Round 4: every stable tile of seeds 2021 .. 2040 (G15_KEEP = 10), run INSIDE a batch of 16
    (the headline's batch size; the other entries are filler tiles)."""
    from lanemapping_amd.pipeline import TilePipeline
    g = golden('g15_e2e_stable.npz')
    net, _ = _g15_net(dev, synth_sd, g)
    seeds = [int(s) for s in g['tile_seeds']]
    assert len(seeds) >= 8, 'the screen of make_golden.py g15 keeps at least 8 stable tiles'
    batch = seeds + [7000 + k for k in range(16 - len(seeds))]
    x = torch.from_numpy(synth.bev_batch(batch, 1152)).to(dev)
    with torch.no_grad():
        o = net({'proj': x})
    c = net.heads._compact
    res = TilePipeline(net).run_batch(x)
    exact = [_check_stable_golden(g, i, i, c, o, res, f'G15 tile {seeds[i]} (entry {i} of a batch of 16)') for i in range(len(seeds))]
    _log_exact('G15', exact, seeds)
    assert sum(exact) >= len(seeds) - _SOFT_BUDGET, f'only {sum(exact)} of {len(seeds)} tiles are identical to the reference in every endpoint'


def test_headline_chain_golden_g17(dev, golden, synth_sd):
    """Golden G17 = the HEADLINE chain against the reference as a chain: seeded 4,194,304-point clouds -> lm_bev_raster_batch (u8 tiles,
    what bench.py times) -> FPN / ViT / head / decode / assembly.  The golden holds what the REFERENCE net produces on u8 / 255 of the C
    oracle's raster of the same clouds (stability-screened like G15): exact vertex set / semantics / endpoints, columns <= 8e-4 px,
    decode outputs within absolute 1e-4."""
    from lanemapping_amd import ops
    from lanemapping_amd.pipeline import TilePipeline
    g = golden('g17_chain.npz')
    net, _ = _g15_net(dev, synth_sd, g)
    seeds = [int(s) for s in g['cloud_seeds']]
    assert len(seeds) >= 2
    clouds = [synth.las_points(s, int(g['n_points'])) for s in seeds]
    offs = np.concatenate([[0], np.cumsum([len(c) for c in clouds])]).tolist()
    points = torch.from_numpy(np.concatenate(clouds)).to(dev)
    kw = {str(k): float(v) for k, v in zip(g['raster_keys'], g['raster_values'])}
    tiles = torch.empty((len(seeds), 1152, 1152, 3), device=dev, dtype=torch.uint8)
    ops.bev_raster_batch(points, offs, [ops.make_raster_params(**kw)] * len(seeds), out_u8=tiles, u8_only=True)
    for i in range(len(seeds)):        # the tile the reference saw (byte sum of the C oracle's raster, stored by make_golden.py g17)
        assert int(tiles[i].cpu().numpy().astype(np.uint64).sum()) == int(g[f'tile_crc{i}']), f'cloud {seeds[i]}: raster differs from the oracle tile'
    with torch.no_grad():
        o = net({'proj': tiles})
    c = net.heads._compact
    res = TilePipeline(net).run_batch(tiles)
    exact = [_check_stable_golden(g, i, i, c, o, res, f'G17 cloud {seeds[i]}') for i in range(len(seeds))]
    _log_exact('G17', exact, seeds)
    assert sum(exact) == len(seeds), f'only {sum(exact)} of {len(seeds)} clouds are identical to the reference in every endpoint'


@pytest.mark.parametrize('B,picks', [(8, (2, 7)), (16, (5, 13))])
def test_tiles_inside_full_batches_vs_oracle(dev, net, synth_sd, B, picks):
    """BASELINE's batch sizes (8 pre-rasterised, 16 fused): two tiles INSIDE a full batch vs the oracle run on those tiles alone -
    raw outputs within 1e-4 of the tensor scale, integer decisions equal wherever the oracle's own margin is >= 1e-4, and the
    batch result equals the single-tile result of the product bit for bit (batch invariance at the real batch sizes)."""
    from oracle import net_ref, decode_ref
    seeds = [6000 + 10 * B + i for i in range(B)]
    x = torch.from_numpy(synth.bev_batch(seeds, 1152)).to(dev)
    cfg = net.cfg
    with torch.no_grad():
        raw = {k: v.clone() for k, v in net.forward_raw({'proj': x}).items()}
        o = net({'proj': x})
    comp = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in net.heads._compact.items()}
    for t in picks:
        xs = torch.from_numpy(synth.bev_batch([seeds[t]], 1152))
        with torch.no_grad():
            ref = net_ref.detector_forward(synth_sd, xs)
            one = net.forward_raw({'proj': xs.to(dev)})
        for k in ('proposal_conf', 'ext2', 'cls2', 'offset2', 'orient', 'semantic_seg', 'endp_est'):
            _close(raw[k][t:t + 1], ref[k], 1e-4, f'B{B} tile {t} {k}')
            assert torch.equal(raw[k][t:t + 1], one[k]), f'B{B} tile {t} {k}: batch result != single-tile result'
        d = decode_ref.decode_column_proposals({k: v.numpy() for k, v in ref.items()})
        e = ref['ext2'].softmax(3)[0]
        ext_margin = torch.minimum((e[..., 1] - e[..., 2]).abs(), (torch.maximum(e[..., 1], e[..., 2]) - cfg.exist_thre).abs()).flatten().numpy()
        ct = torch.topk(ref['cls2'], 2, dim=-1).values[0]
        cls_margin = (ct[..., 0] - ct[..., 1]).flatten().numpy()
        for mine, want, margin, name in ((comp['prop_v_ext'][t].cpu().numpy(), d['prop_v_ext'][0].numpy(), ext_margin, 'prop_v_ext'),
                                         (comp['cls_idx'][t].cpu().numpy(), d['cls_idx'][0].numpy(), cls_margin, 'cls_idx')):
            bad = np.flatnonzero(np.asarray(mine).reshape(-1) != np.asarray(want).reshape(-1))
            assert np.all(margin[bad] < 1e-4) and bad.size <= 4, f'B{B} tile {t} {name}: {bad.size} mismatches'
        assert np.array_equal(np.stack(np.nonzero(o['endp'][t].numpy()), 1), np.stack(np.nonzero(d['endp'][0].numpy()), 1))


def test_rowref_pipeline_graph_replay_bit_identical(dev):
    """Config 4 (RowRef head: device-side lane selection, masked attention on the fixed token grid) through TilePipeline(use_graph=True):
    captured and replayed it gives the same lanes and endpoints, bit for bit, as the eager launches - on the batch it was captured with and
    on different tiles through the same graph (bench.py switches graphs on by itself at <= 4 host cores per rank)."""
    from lanemapping_amd.boundary import build_net_from_config
    from lanemapping_amd.pipeline import TilePipeline
    net = build_net_from_config('Proj28_GFC-T3_RowRef_82_73_laser', device='cpu')
    synth.fill_module_(net, 2021)
    net = net.to(dev)
    eager, graph = TilePipeline(net, use_graph=False), TilePipeline(net, use_graph=True)
    for seeds in ([2021, 2022], [2030, 2031], [2040, 2041]):
        x = torch.from_numpy(synth.bev_batch(seeds, 1152)).to(dev)
        want = eager.run_batch(x)
        got = graph.run_batch(x)
        assert len(want) == len(got) == len(seeds)
        for (la, ea), (lb, eb) in zip(want, got):
            assert np.array_equal(np.asarray(la), np.asarray(lb)) and np.array_equal(np.asarray(ea), np.asarray(eb))
    assert len(graph._graphs) == 1


@pytest.mark.parametrize('seed', [3001, 3002])
def test_full_tiles_other_seeds_vs_oracle(dev, net, synth_sd, seed):
    """Full 1152^2 tiles the goldens do not cover: raw outputs within 1e-4 of the tensor scale, and every integer decision
    (existence class, orientation, column bin, semantic class) equal to the oracle's wherever the ORACLE's own decision
    margin is >= 1e-4 (the oracle is bit-identical to the reference on the goldens)."""
    from oracle import net_ref, decode_ref
    x = torch.from_numpy(synth.bev_batch([seed], 1152))
    cfg = net.cfg
    with torch.no_grad():
        ref = net_ref.detector_forward(synth_sd, x)
        raw = net.forward_raw({'proj': x.to(dev)})
        for k in ('proposal_conf', 'ext2', 'cls2', 'offset2', 'orient', 'semantic_seg', 'endp_est'):
            _close(raw[k], ref[k], 1e-4, k)
        o = net({'proj': x.to(dev)})
    d = decode_ref.decode_column_proposals({k: v.numpy() for k, v in ref.items()})
    sm = ref['semantic_seg'].softmax(1)[0]
    sem_margin = torch.minimum((sm[1] - sm[2]).abs(), (torch.maximum(sm[1], sm[2]) - cfg.coor_thre).abs()).flatten()
    e = ref['ext2'].softmax(3)[0]
    ext_margin = torch.minimum((e[..., 1] - e[..., 2]).abs(), (torch.maximum(e[..., 1], e[..., 2]) - cfg.exist_thre).abs()).flatten()
    ot = torch.topk(ref['orient'], 2, dim=1).values[0]
    ct = torch.topk(ref['cls2'], 2, dim=-1).values[0]

    def outside_noise(mine, want, margin, name):
        bad = np.flatnonzero(np.asarray(mine).reshape(-1) != np.asarray(want).reshape(-1))
        assert np.all(margin.numpy()[bad] < 1e-4), f'{name}: mismatch where the oracle margin is >= 1e-4'
        return bad.size
    n1 = outside_noise(o['prop_v_ext'][0].numpy(), d['prop_v_ext'][0].numpy(), ext_margin, 'prop_v_ext')
    n2 = outside_noise(o['orient'][0].numpy(), d['orient'][0].numpy(), (ot[0] - ot[1]).flatten(), 'orient')
    n3 = outside_noise(o['semantic_seg'][0].numpy(), d['semantic_seg'][0].numpy(), sem_margin, 'semantic_seg')
    n4 = outside_noise(net.heads._compact['cls_idx'][0].cpu().numpy(), d['cls_idx'][0].numpy(), (ct[..., 0] - ct[..., 1]).flatten(), 'cls_idx')
    print(f'seed {seed}: flips inside the noise margin: ext {n1}, orient {n2}, semantic {n3}, column bin {n4}')
    assert n1 + n2 + n4 <= 4 and n3 <= 64


def test_tile_pipeline_graph_replay_bit_identical(dev):
    """TilePipeline(use_graph=True): the device part of a batch captured into one HIP graph and replayed gives the same polylines and
    endpoints, bit for bit, as launching the same kernels one by one - on the batch it was captured with, on different tiles through the
    same graph (static input buffer), and for a second batch shape (second graph)."""
    from lanemapping_amd.boundary import build_net_from_config
    from lanemapping_amd.pipeline import TilePipeline
    net = build_net_from_config('Proj_polyline_fpn_vit_vertex_2', device='cpu')
    synth.fill_module_(net, 2021)
    net = net.to(dev)
    eager, graph = TilePipeline(net, use_graph=False), TilePipeline(net, use_graph=True)
    for seeds in ([2021, 2022], [2030, 2031], [2040]):
        x = torch.from_numpy(synth.bev_batch(seeds, 1152)).to(dev)
        want = eager.run_batch(x)
        got = graph.run_batch(x)
        assert len(want) == len(got) == len(seeds)
        for (la, ea), (lb, eb) in zip(want, got):
            assert np.array_equal(np.asarray(la), np.asarray(lb)) and np.array_equal(np.asarray(ea), np.asarray(eb))
    assert len(graph._graphs) == 2
    # a graph bakes the packed-weight pointers in: new weights (load_ckpt / load_state_dict / in-place edits) must force a recapture,
    # never a silent replay of the old ones
    synth.fill_module_(net, 77)
    x = torch.from_numpy(synth.bev_batch([2021, 2022], 1152)).to(dev)
    want, got = eager.run_batch(x), graph.run_batch(x)
    for (la, ea), (lb, eb) in zip(want, got):
        assert np.array_equal(np.asarray(la), np.asarray(lb)) and np.array_equal(np.asarray(ea), np.asarray(eb))
    # the cache is bounded (every graph pins a private activation pool)
    for n in (1, 3, 4):
        graph.run_batch(torch.from_numpy(synth.bev_batch([5 + i for i in range(n)], 1152)).to(dev))
    assert len(graph._graphs) <= TilePipeline.MAX_GRAPHS
    graph.clear_graphs()
    assert len(graph._graphs) == 0


@pytest.mark.parametrize('switch', ['LANEMAP_WINO_F44=0', 'LANEMAP_WINO_SPLIT=1', 'LANEMAP_GRAPHS=1', 'LANEMAP_MERGE_BRANCH_CONVS=0',
                                    'LANEMAP_W44_ORDER=0 LM_CONV_LATERAL=0 LM_CONV_TINYK=0 LM_CONV_SMALLM=100 LANEMAP_WINO_F44_MIN_CIN=128 LM_RASTER_BAND_ROWS=16 LANEMAP_ROCTX=1',
                                    'LM_STEM_VALU=1 LM_GN_UP_LDS=0 LM_GN_SUM_LDS=0 LM_SMALL_CONV_VALU=1 LM_HEAD_TOKENS_GATHER=1 LM_HEAD_STAGE2_DIRECT=1'])
def test_goldens_under_every_advertised_switch(switch):
    """README's runtime switches are read once per process, so each non-default setting gets its own interpreter: the end-to-end
    goldens (G10: one full tile against the reference's outputs, margin-aware; G15: the stability-screened tiles whose final
    polylines must equal the reference's; G17: the headline chain from 4.19 M-point clouds) run under it.  A switch that is not tested here does not exist."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ)
    for kv in switch.split():
        k, v = kv.split('=')
        env[k] = v
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_gpu_2_goldens.py'), '-x', '-q', '-m', 'gpu', '-k',
                        'test_end_to_end_golden_g10 or test_end_to_end_stable_golden_g15 or test_headline_chain_golden_g17 or test_tile_pipeline_graph_replay'],
                       capture_output=True, text=True, timeout=1500, cwd=root, env=env)
    assert r.returncode == 0 and ' passed' in r.stdout, (r.stdout + r.stderr)[-3000:]
