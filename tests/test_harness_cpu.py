"""CPU: the entry-point contract of the harness and the evaluation loop's bookkeeping (f3).

  * the reference's four unmodified configs/Proj_*.py load through the boundary and give the repo configs' state-dict layout
    (skipped where /root/reference does not exist, i.e. on the GPU box);
  * the dataset listing and the evaluation ground truth equal golden G18 (the imported reference's LaserLaneProposal on a synthetic
    <data_root> that cases.write_dataset re-creates here);
  * the test loop's counter sums and P / R / F1 formulas equal golden G19;
  * Runner's entry points carry the reference's signatures, refuse unknown keywords and refuse an empty tile list.
"""
import inspect
import os

import numpy as np
import pytest

import cases
from lanemapping_amd import datasets, metric_utils
from lanemapping_amd.boundary import load_config, build_net_from_config

REF = '/root/reference'
CONFIGS = ('Proj_polyline_fpn_vit_vertex_2', 'Proj_FPN_Seg', 'Proj28_GFC-T3_RowRef_82_73_laser', 'Proj_polyline_lidarconv_vit_vertex_2')


@pytest.mark.skipif(not os.path.isdir(REF), reason='the reference tree only exists in the build container')
@pytest.mark.parametrize('name', CONFIGS)
def test_unmodified_reference_configs_build_the_same_layout(name):
    ref_net = build_net_from_config(f'{REF}/configs/{name}.py', device='cpu')
    own_net = build_net_from_config(name, device='cpu')
    a = {k: tuple(v.shape) for k, v in ref_net.state_dict().items()}
    b = {k: tuple(v.shape) for k, v in own_net.state_dict().items()}
    assert a == b and list(a) == list(b) and len(a) > 300
    rc, oc = load_config(f'{REF}/configs/{name}.py'), load_config(name)
    # the keys the entry points read (runner.py:57-66, :690-697, :755) agree between the trimmed configs and the published ones
    assert dict(rc.dataset.test) == dict(oc.dataset.test) and rc.dataset.train.type == oc.dataset.train.type
    assert (rc.log_dir, rc.batch_size, rc.seed, rc.validate_buffer) == (oc.log_dir, oc.batch_size, oc.seed, oc.validate_buffer)


@pytest.fixture(scope='module')
def g18_root(tmp_path_factory, golden):
    g = golden('g18_dataset.npz')
    root = tmp_path_factory.mktemp('g18')
    cases.write_dataset(str(root), n_tiles=int(g['n_tiles']), seed=int(g['seed']))
    return str(root), g


def test_dataset_listing_matches_reference(g18_root):
    root, g = g18_root
    for mode in ('test', 'valid', 'single', 'all', 'infer_only', 'train'):
        ents = datasets.load_datadir(root, 'data_split-shuffle.json', mode, shuffle_seed=int(g['shuffle_seed']))
        # the reference's stems keep the '.' of '<stem>.json' (laserlane_proposals.py:534); names are cut to 11 characters either way
        assert [e['stem'] + '.' for e in ents] == list(g['stems_' + mode]), mode
        assert [os.path.relpath(e['image'], root) for e in ents] == list(g['images_' + mode]), mode
    ents = datasets.load_datadir(root, 'data_split-shuffle.json', 'test', shuffle_seed=int(g['shuffle_seed']))
    for key in ('seq', 'mask', 'instance', 'endp'):
        assert [os.path.relpath(e[key], root) for e in ents] == list(g[key + '_test'])
    unshuffled = [e['stem'] for e in datasets.load_datadir(root, 'data_split-shuffle.json', 'test')]
    assert unshuffled == cases.dataset_stems(int(g['n_tiles'])) and unshuffled != [e['stem'] for e in ents]


def test_split_entries_follow_the_dataset_classes(g18_root):
    root, _ = g18_root
    cfg = load_config('Proj_polyline_fpn_vit_vertex_2')
    split = dict(cfg.dataset.test, data_root=root)
    ents = datasets.split_entries(split, cfg)
    assert len(ents) == 7 and ents[0]['image'].startswith(os.path.join(root, 'cropped_tiff'))
    assert len(datasets.split_entries(dict(split, mode='infer_only'), cfg)) == 6            # the 'pretrain' list
    with pytest.raises(AssertionError):
        datasets.split_entries(dict(split, mode='val'), cfg)                                 # C12: the configs' `val` split never worked
    with pytest.raises(AssertionError):
        datasets.split_entries(dict(split, type='LaserLane', mode='infer_only'), cfg)        # laserlane.py:34 has no infer_only
    assert len(datasets.split_entries({'type': 'LaserLane', 'data_root': root, 'mode': 'test'}, cfg)) == 7
    with pytest.raises(KeyError):
        datasets.split_entries(dict(split, type='NoSuchDataset'), cfg)


def test_evaluation_ground_truth_matches_reference(g18_root):
    root, g = g18_root
    cfg = load_config('Proj_polyline_fpn_vit_vertex_2')
    ents = datasets.load_datadir(root, 'data_split-shuffle.json', 'test', shuffle_seed=int(g['shuffle_seed']))
    merged = 0
    for i, e in enumerate(ents):
        gt = datasets.load_eval_gt(e, cfg)
        assert gt['lc_coor_raw'].dtype == np.float32 and np.array_equal(gt['lc_coor_raw'], g[f'lc_coor_raw_{i}'])
        m = gt['mask']
        nz = np.concatenate([np.argwhere(m != 0), m[m != 0][:, None]], axis=1).astype(np.int32)
        assert np.array_equal(nz, g[f'mask_nz_{i}'])
        assert np.array_equal(np.argwhere(gt['endp_map'] > 0).astype(np.int32), g[f'endp_nz_{i}'])
        assert np.array_equal(gt['endp_map'][gt['endp_map'] > 0], g[f'endp_val_{i}'])
        unmerged = datasets.load_eval_gt(e, cfg, merge_connect_lines=False)['lc_coor_raw']
        merged += int(not np.array_equal(unmerged, gt['lc_coor_raw']))
        assert np.array_equal(datasets.klane_coor_label(gt['label_raw'], 12), unmerged)     # same columns, px units, no merge
    assert merged >= 3                        # the connected pair of cases.label_case is really exercised


def test_eval_loop_accumulation_matches_reference(golden):
    from lanemapping_amd.runner import _prf
    g = golden('g19_eval_loop.npz')
    buf = int(g['validate_buffer'])
    tot = np.zeros(8)
    for seed in g['seeds']:
        label, pred, egt, epr = cases.metric_case(int(seed))
        tot[0:4] += metric_utils.cal_coor_measures(label, pred, 'conf', offset_thre=buf)[3:7]
        tot[4:8] += metric_utils.eval_metric_endp_detector(epr, egt, r_thre=buf * 2)[3:7]
    assert np.array_equal(tot, g['counters'])
    assert np.array_equal(np.array(_prf(*tot[0:4]) + _prf(*tot[4:8])), g['prf'])           # same operations, same doubles
    assert _prf(0, 0, 0, 0) == (0., 0., 0.)


def test_runner_entry_points_have_the_reference_signatures():
    from lanemapping_amd.runner import Runner, load_config_and_runner

    def positional(fn):
        return [(p.name, p.default) for p in inspect.signature(fn).parameters.values()
                if p.kind == p.POSITIONAL_OR_KEYWORD and p.name != 'self']
    # baseline/engine/runner.py:690-692, :606, :945-948, :57
    assert positional(Runner.infer_lane_coordinate_endpoint_semantics) == [
        ('path_ckpt', None), ('mode_data', None), ('mode_view', False), ('gt_avail', True), ('write_lane_vertex', False),
        ('eval_coor', True), ('eval_endp', True), ('eval_semantic', True)]
    assert positional(Runner.infer_lane_coordinate) == [('path_ckpt', None), ('mode_view', False), ('gt_avail', True),
                                                        ('write_lane_vertex', False)]
    assert positional(Runner.infer_lane_geometry_segmentation_segmentor) == [('path_ckpt', None), ('mode_view', False),
                                                                             ('write_lane_vertex', False)]
    assert [n for n, _ in positional(load_config_and_runner)] == ['path_config', 'gpus']
    for fn in (Runner.infer_lane_coordinate_endpoint_semantics, Runner.infer_lane_coordinate,
               Runner.infer_lane_geometry_segmentation_segmentor):
        assert not any(p.kind == p.VAR_KEYWORD for p in inspect.signature(fn).parameters.values())     # no **_ignored


def test_load_config_and_runner_sets_the_reference_directories(tmp_path, monkeypatch):
    import lanemapping_amd.runner as R
    monkeypatch.setattr(R, 'Runner', lambda cfg: ('runner', cfg))
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'configs', 'Proj_polyline_fpn_vit_vertex_2.py')).read()
    p = tmp_path / 'configs_Proj_polyline_fpn_vit_vertex_2.py'
    p.write_text(src.replace("log_dir = './logs'", f"log_dir = {str(tmp_path / 'logs')!r}"))
    cfg, runner = R.load_config_and_runner(str(p), '0')
    assert cfg.log_dir == str(tmp_path / 'logs') + '/vis' and cfg.work_dirs == cfg.log_dir + '/LaserLaneProposal' and cfg.gpus == 1
    assert os.path.isdir(cfg.work_dirs) and runner == ('runner', cfg)
    # several ids: the same directories, cfg.gpus = their number, and a MultiGpuRunner (one fresh process per id; both on device 0 under the test hook)
    monkeypatch.setenv('LANEMAP_TEST_DEVICE', '0')
    cfg, runner = R.load_config_and_runner(str(p), '0,1,2')
    assert cfg.work_dirs == str(tmp_path / 'logs') + '/vis/LaserLaneProposal' and cfg.gpus == 3 and runner.gpu_ids == [0, 1, 2]


def test_runner_refuses_what_it_cannot_honour(g18_root, tmp_path):
    from lanemapping_amd.runner import Runner
    root, _ = g18_root
    cfg = load_config('Proj_polyline_fpn_vit_vertex_2')
    r = Runner.__new__(Runner)                                   # no GPU here: the listing / argument checks come before any device work
    r.cfg = cfg
    with pytest.raises(TypeError):
        r.infer_lane_coordinate_endpoint_semantics(mode_data=cfg.dataset.test, no_such_keyword=1)
    empty = tmp_path / 'empty'
    empty.mkdir()
    with pytest.raises(ValueError, match='no tiles'):
        r.infer_lane_coordinate_endpoint_semantics(tiles=str(empty), work_dirs=str(tmp_path / 'o'))
    with pytest.raises(FileNotFoundError):                       # the published relative data_root does not exist here
        r.infer_lane_coordinate_endpoint_semantics(mode_data=cfg.dataset.test, gt_avail=False)
    ents = r._entries(dict(cfg.dataset.test, data_root=root), None)
    assert [n for n, _, _ in ents] == sorted(s[0:11] for s in cases.dataset_stems(7)) and all(len(n) == 11 for n, _, _ in ents)
    with open(os.path.join(root, 'data_split-empty.json'), 'w') as f:
        f.write('{"train": [], "test": [], "valid": [], "single": [], "pretrain": []}')
    with pytest.raises(ValueError, match='no tiles'):
        r._entries(dict(cfg.dataset.test, data_root=root, data_split_file='data_split-empty.json'), None)


def test_stage_stats_tool_on_synthetic_traces(tmp_path):
    """tools/r5/stage_stats.py on a hand-made rocprofv3 output directory (kernel / marker / HIP-API traces in the CSV layout of rocprofv3
    1.x on the GPU boxes): only kernels that START inside the `timed_steps` range are counted, each is attributed to the innermost roctx
    range open on the launching thread at its launch call (matched through the correlation id), copies and ATen kernels are tallied."""
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    d = tmp_path / 'trace'
    d.mkdir()
    (d / 'p_marker_api_trace.csv').write_text(
        '"Domain","Function","Process_Id","Thread_Id","Correlation_Id","Start_Timestamp","End_Timestamp"\n'
        '"MARKER_CORE_RANGE_API","pcencoder",1,7,100,500,900\n'             # warm-up batch: before the timed range
        '"MARKER_CORE_RANGE_API","timed_steps",1,7,101,1000,9000\n'
        '"MARKER_CORE_RANGE_API","pcencoder",1,7,102,1100,5000\n'
        '"MARKER_CORE_RANGE_API","fpn.layer1",1,7,103,1200,2000\n'
        '"MARKER_CORE_RANGE_API","decode",1,7,104,5100,5200\n')
    (d / 'p_hip_api_trace.csv').write_text(
        '"Domain","Function","Process_Id","Thread_Id","Correlation_Id","Start_Timestamp","End_Timestamp"\n'
        '"HIP_RUNTIME_API_EXT","hipLaunchKernel",1,7,1,600,610\n'
        '"HIP_RUNTIME_API_EXT","hipLaunchKernel",1,7,2,1300,1310\n'
        '"HIP_RUNTIME_API_EXT","hipLaunchKernel",1,7,3,3000,3010\n'
        '"HIP_RUNTIME_API_EXT","hipMemcpyAsync",1,7,4,5150,5160\n'
        '"HIP_RUNTIME_API_EXT","hipLaunchKernel",1,7,5,5150,5160\n')
    hdr = ('"Kind","Agent_Id","Queue_Id","Stream_Id","Thread_Id","Dispatch_Id","Kernel_Id","Kernel_Name","Correlation_Id","Start_Timestamp",'
           '"End_Timestamp","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Workgroup_Size_X","Workgroup_Size_Y",'
           '"Workgroup_Size_Z","Grid_Size_X","Grid_Size_Y","Grid_Size_Z"\n')
    row = '"KERNEL_DISPATCH","Agent 2",1,0,7,{i},8,"{n}",{c},{s},{e},0,0,8,0,32,256,1,1,256,1,1\n'
    (d / 'p_kernel_trace.csv').write_text(
        hdr + row.format(i=1, n='(anonymous namespace)::wino44_kernel((anonymous namespace)::W44Params)', c=1, s=700, e=800)       # warm-up: not counted
        + row.format(i=2, n='(anonymous namespace)::wino44_kernel((anonymous namespace)::W44Params)', c=2, s=1400, e=2400)
        + row.format(i=3, n='void (anonymous namespace)::conv_mfma_kernel<64, 64, 32, 32, false, 1>((anonymous namespace)::ConvParams)', c=3, s=3100, e=3400)
        + row.format(i=4, n='__amd_rocclr_copyBuffer', c=4, s=5170, e=5180)
        + row.format(i=5, n='(anonymous namespace)::decode_orient_kernel(float const*, int, int, unsigned char*, long)', c=5, s=5190, e=5200))
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'r5', 'stage_stats.py'), str(d), '2'], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    out = r.stdout
    assert '4 kernel dispatches started inside the timed_steps range' in out and '1 dispatches outside it are not counted' in out
    assert 'copy kernels 0.5 per step, ATen kernels 0.0 per step' in out
    lines = {l.split()[0]: l for l in out.splitlines() if l and not l.startswith('#')}
    assert 'wino44_kernel(W44Params)' in lines and ' 0.5/step' in lines['wino44_kernel(W44Params)']       # the warm-up launch is not in it
    import re
    stage = {m.group(1).strip(): float(m.group(2)) for m in re.finditer(r'^(.*?)\s+([0-9.]+) launches/step', out.split('## per stage')[1], re.M)}
    assert stage['fpn.layer1'] == 0.5 and stage['pcencoder'] == 0.5 and stage['decode'] == 0.5      # innermost range wins; launches per step
    assert stage['(no launch record)'] == 0.5                                                          # the copy has no Launch record


def test_trace_ranges_are_free_when_off():
    """lanemapping_amd.trace: without LANEMAP_ROCTX the stage ranges do not load any library."""
    import importlib
    from lanemapping_amd import trace
    importlib.reload(trace)
    assert trace.ENABLED == (os.environ.get('LANEMAP_ROCTX', '0') != '0')
    if not trace.ENABLED:
        with trace.stage('x'):
            trace.push('y'); trace.pop(); trace.mark('z')
        assert trace._lib is None


# ------------------------------------------------------------------------------------------------ GPUS_EN = '0,1,...' (runner_ranks.py)
def _config_copy(tmp_path, name='Proj_polyline_fpn_vit_vertex_2'):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, 'configs', name + '.py')).read().replace("log_dir = './logs'", f"log_dir = {str(tmp_path / 'logs')!r}")
    path = tmp_path / ('configs_' + name + '.py')
    path.write_text(src)
    return str(path)


def test_gpus_string_is_parsed_like_the_reference_or_refused(tmp_path, monkeypatch):
    """test_gpu_0.py:7-9 / runner.py:57-66: `gpus` is a comma-separated id list and cfg.gpus its length.  A malformed list raises
    ValueError before anything is built; more ids than visible GPUs (none in this container) raises RuntimeError that names the
    torch.distributed.run command - a multi-id string is never silently served by one GPU."""
    from lanemapping_amd import runner_ranks
    from lanemapping_amd.runner import load_config_and_runner
    assert runner_ranks.parse_gpus('0') == [0] and runner_ranks.parse_gpus('0, 1,2') == [0, 1, 2] and runner_ranks.parse_gpus(3) == [3]
    for bad in ('', '0,', 'a', '0,0', '0,-1', '0;1'):
        with pytest.raises(ValueError, match='gpus='):
            load_config_and_runner(_config_copy(tmp_path), bad)
    monkeypatch.delenv('LANEMAP_TEST_DEVICE', raising=False)
    import torch
    if torch.cuda.device_count() < 2:
        with pytest.raises(RuntimeError, match=r'names 2 GPUs.*torch\.distributed\.run --nnodes=1 --nproc-per-node 2'):
            load_config_and_runner(_config_copy(tmp_path), '0,1')
    # which device every rank drives: the reference's masked list -> visible devices 0..n-1 (DataParallel's range(cfg.gpus)); no mask -> the ids
    monkeypatch.delenv('HIP_VISIBLE_DEVICES', raising=False)
    monkeypatch.delenv('CUDA_VISIBLE_DEVICES', raising=False)
    assert runner_ranks.rank_devices([2, 3]) == ([2, 3], 'nccl')
    monkeypatch.setenv('CUDA_VISIBLE_DEVICES', '2,3')
    assert runner_ranks.rank_devices([2, 3]) == ([0, 1], 'nccl')
    monkeypatch.setenv('HIP_VISIBLE_DEVICES', '4,5,6')
    assert runner_ranks.rank_devices([1, 2]) == ([1, 2], 'nccl')
    monkeypatch.setenv('LANEMAP_TEST_DEVICE', '0')
    assert runner_ranks.rank_devices([0, 1, 2]) == ([0, 0, 0], 'gloo')


def test_multi_gpu_runner_facade_and_rank_failure(tmp_path, monkeypatch, synth_sd):
    """MultiGpuRunner carries the Runner entry signatures, keeps a CPU net that load_ckpt fills strictly, ships a picklable job
    (config tree as plain dicts), and a failing rank surfaces as RuntimeError with that rank's stderr (here: no GPU in the build
    container, so every fresh rank process refuses to start) instead of a hang or a silent single-GPU run."""
    import pickle
    import torch
    from lanemapping_amd import runner_ranks
    from lanemapping_amd.config import Config
    from lanemapping_amd.runner import Runner, load_config_and_runner
    monkeypatch.setenv('LANEMAP_TEST_DEVICE', '0')
    cfg, r = load_config_and_runner(_config_copy(tmp_path), '0,1')
    assert isinstance(r, runner_ranks.MultiGpuRunner) and cfg.gpus == 2 and r.devices == [0, 0] and r.backend == 'gloo'
    for name in ('infer_lane_coordinate_endpoint_semantics', 'infer_lane_coordinate', 'infer_lane_geometry_segmentation_segmentor'):
        assert inspect.signature(getattr(r, name)) == inspect.signature(getattr(Runner(cfg, device='cpu'), name)), name
    path_ckpt = str(tmp_path / 'best.pth')
    torch.save({'net': {'module.' + k: v for k, v in synth_sd.items()}}, path_ckpt)
    r.load_ckpt(path_ckpt)
    assert all(torch.equal(v, synth_sd[k]) for k, v in r.net.state_dict().items())
    assert all(p.device.type == 'cpu' for p in r.net.parameters())
    torch.save({'net': {k: v for k, v in list(synth_sd.items())[1:]}}, path_ckpt)
    with pytest.raises(RuntimeError, match='Missing key'):
        r.load_ckpt(path_ckpt)
    plain = pickle.loads(pickle.dumps(runner_ranks._plain(cfg)))
    assert type(plain) is dict and type(plain['dataset']['test']) is dict and Config(plain) == cfg
    assert Config(plain).dataset.test.mode == cfg.dataset.test.mode
    with pytest.raises(NotImplementedError, match='single-GPU'):
        r.infer_las_to_map([])
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match=r'2 of 2 GPU ranks failed(.|\n)*rank 1 \(GPU 1, exit code 1\)(.|\n)*GPU index 0 is not visible'):
            r.infer_lane_coordinate_endpoint_semantics(tiles=str(tmp_path), write_lane_vertex=True)
