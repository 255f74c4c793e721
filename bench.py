#!/usr/bin/env python3
"""Throughput bench of the lane-mapping hot path on MI355X (contract: see DESIGN.md §Measurement).

    python bench.py [--gpus N --steps K --warmup W]          (N > 1 without torchrun: spawns the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Default workload = BASELINE.json's headline, configs[2]: on-GPU LAS -> BEV raster of 16 x 4,194,304 points resident in HBM +
configs/Proj_polyline_fpn_vit_vertex_2.py at batch 16, all the way to lane polylines: raster -> FPN -> ViT -> column-proposal
head -> decode (GPU), endpoint clustering + polyline assembly (host C++ threads, overlapped), plus one all-gather of the
fixed-shape polyline blocks when N > 1.  `--workload tiles` = configs[1] (pre-rasterised, batch 8), `rowref` = configs[3]
(RowRef head, batch 8), `lidar` = configs[4] (sparse-conv encoder).  Rank 0 prints ONE JSON line.
"""
import argparse
import datetime
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_F32_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 MFMA peak
MFMA_F16_PEAK_TFLOPS = 2500.0     # same guide: dense fp16 / bf16 MFMA peak (the split second line is priced against it; its
                                  # v_mfma_f32_32x32x8_f16 instruction alone peaks at half of that)
HBM_PEAK_GBS = 8000.0             # same guide: HBM3E 8 TB/s
BATCH = 8
N_PTS = 4194304                   # points per tile of the fused workload
PMC_FILE = os.path.join(ROOT, 'profiles', 'pmc_traffic.json')     # counter bytes per step, collected offline (tools/pmc_traffic.sh)


def csrc_sha16():
    """Stamp of every source of the library (lanemapping_amd/csrc, device and host): the GPU-suite logs carry it."""
    import hashlib
    d = os.path.join(ROOT, 'lanemapping_amd', 'csrc')
    h = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith(('.hip', '.cpp', '.h')):
            h.update(name.encode())
            with open(os.path.join(d, name), 'rb') as f:
                h.update(f.read())
    return h.hexdigest()[:16]


def kernel_sha16():
    """Stamp of the DEVICE sources (the .hip files and the headers they include): tools/pmc_traffic.sh stores it with the counter bytes it
    collects, bench.py compares - `traffic_stale` says the committed counters were measured on other kernels than the ones that just ran
    (a change of the host-only sources - PNG reader, polyline assembly, JSON writer - cannot move a kernel's HBM traffic)."""
    import hashlib
    import re
    d = os.path.join(ROOT, 'lanemapping_amd', 'csrc')
    names = sorted(n for n in os.listdir(d) if n.endswith('.hip'))
    headers = set()
    for n in names:
        with open(os.path.join(d, n), 'r') as f:
            headers.update(re.findall(r'^\s*#include\s+"([^"]+)"', f.read(), flags=re.M))
    h = hashlib.sha256()
    for name in names + sorted(x for x in headers if os.path.exists(os.path.join(d, x))):
        h.update(name.encode())
        with open(os.path.join(d, name), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def usable_cores():
    """Cores this process may actually use: affinity mask, capped by the cgroup CPU quota if there is one."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, n)


def cpu_baseline(budget_s=25.0, max_threads=64):
    """Oracle ("port" of the reference's CPU path: torch-CPU fp32 net + NumPy decode/post-proc) timed on this host's cores on a
    bounded sample of the same workload: batch 1 for about half of `budget_s`, then ONE batch of 8 (SURVEY §8d asks for both)."""
    sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
    from lanemapping_amd import synth
    from lanemapping_amd.boundary import build_net_from_config
    from oracle import net_ref, decode_ref, postproc_ref
    cores = min(usable_cores(), max_threads)
    torch.set_num_threads(cores)
    net = build_net_from_config('Proj_polyline_fpn_vit_vertex_2', device='cpu')     # the CPU baseline is always the config-2 chain
    synth.fill_module_(net, 2021)
    sd = {k: v for k, v in net.state_dict().items()}

    def run(seeds):
        x = torch.from_numpy(synth.bev_batch(seeds, 1152))
        t = time.perf_counter()
        with torch.no_grad():
            raw = net_ref.detector_forward(sd, x)
        d = decode_ref.decode_column_proposals({k: v.numpy() for k, v in raw.items()})
        for b in range(len(seeds)):
            postproc_ref.assemble_tile(d['prop_conf'][b, :, 1].numpy(), d['prop_v_ext'][b].numpy(), d['cls_offset'][b].numpy(),
                                       d['bi_seg'][b].numpy(), d['endp'][b].numpy())
        return time.perf_counter() - t

    warm = run([2021])                          # warm-up tile (also sizes the sample)
    n = int(max(1, min(8, (budget_s * 0.5) // max(warm, 1e-3))))
    times = [run([2022 + i]) for i in range(n)] if warm < budget_s else [warm]
    dt = sum(times)
    out = {'value': len(times) / dt, 'unit': 'tiles/s', 'cores': cores, 'kind': 'port',
           'sample': f'{len(times)} synthetic 1152x1152 tiles, batch 1, oracle net_ref+decode_ref+postproc_ref (tile '
                     f'generation excluded), torch {torch.get_num_threads()} threads of {os.cpu_count()} logical CPUs, '
                     f'after 1 warm-up tile ({warm:.1f} s)'}
    if warm * 8 < budget_s * 1.5:               # one batch of 8 when it fits the budget (about 8 x the batch-1 tile time)
        t8 = run([2030 + i for i in range(8)])
        out['batch8'] = {'value': 8 / t8, 'unit': 'tiles/s', 'sample': 'one batch of 8 tiles through the same chain'}
    return out


SPLIT_LINE = os.environ.get('LANEMAP_WINO_SPLIT', '0') != '0'        # this process IS a second-line run


def second_line(args):
    """The declared second line: this command once more in a CHILD process (the switch is read at import) under LANEMAP_WINO_SPLIT=1 - the
    F(4x4) Winograd products on the fp16 matrix pipe, every fp32 operand in two fp16 terms, three products, fp32 accumulation
    (csrc/conv_wino44.hip w44_split2).  fp32-accurate, not bit-identical; held to the same goldens by the every-switch test.  Never `value`."""
    import subprocess
    steps = max(10, min(args.steps, 40))
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', str(steps), '--warmup', '3', '--no-cpu-baseline', '--no-second-line']
    if args.no_graphs:
        cmd.append('--no-graphs')
    if args.streams is not None:
        cmd += ['--streams', str(args.streams)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=ROOT, env=dict(os.environ, LANEMAP_WINO_SPLIT='1'))
        lines = [l for l in r.stdout.strip().split('\n') if l.startswith('{')]
        if r.returncode != 0 or not lines:
            return {'error': f'second-line run failed (exit code {r.returncode}): ' + (r.stderr or r.stdout)[-400:]}
        d = json.loads(lines[-1])
    except Exception as e:          # the headline must print whatever happens to the second line
        return {'error': f'second-line run failed: {e!r}'}
    rf = d['roofline']
    return {'what': 'the same command under LANEMAP_WINO_SPLIT=1: Winograd F(4x4) products as fp16 x 2 split terms (three v_mfma_f32_32x32x8_f16 per '
                    'fp32 product pair, fp32 accumulation), fp32-accurate but NOT bit-identical to the exact path; a second line, never the headline',
            'value': d['value'], 'unit': d['unit'], 'ms_per_step': d['ms_per_step'], 'steps': d['steps'], 'dtype': d['dtype'],
            'windows_tiles_per_s': d['config'].get('windows_tiles_per_s'), 'stream_check': d['config'].get('stream_check'),
            'raster_check': d['config'].get('raster_check'),
            'roofline': {'bound': 'mfma', 'per_kernel': rf['per_kernel'], 'kernel_ms_per_step': rf['kernel_ms_per_step'],
                         'winograd_ms_per_step': rf['winograd_ms_per_step'], 'frac': rf['frac'], 'scope': rf['scope'],
                         'note': 'frac = time-weighted executed MFMA FLOPs / peak of each launch\'s dtype: the split Winograd launches execute 3 x '
                                 'the fp32 kernel\'s products as fp16 MFMAs and are priced against the 2.5 PFLOP/s dense fp16 peak, the rest against '
                                 'the fp32 MFMA peak'},
            'parity': 'tests/test_gpu_1_kernels.py::test_conv_winograd44_split_second_line (bit-identical to its twin, error vs fp64 like the exact '
                      'kernel), tests/test_gpu_2_goldens.py::test_goldens_under_every_advertised_switch[LANEMAP_WINO_SPLIT=1] (G10 / G15 / G17)'}


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as child processes (one per GPU) BEFORE this process
    touches the GPU, relay rank 0's JSON line through the inherited stdout and exit with the worst child's code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, 'bench.py')] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    codes = [p.wait() for p in procs]
    sys.exit(max(abs(c) for c in codes))


def gpu_numa_topology():
    """NUMA node of every GPU of this node WITHOUT touching HIP, and the CPUs of every NUMA node (the reference pins nothing:
    baseline/engine/runner.py:89-104 leaves DataParallel's workers wherever the scheduler puts them).  GPUs = the amdgpu DRM cards in PCI
    bus order - the order ROCr enumerates them in (assumed, not checked: the chosen node and cores are printed in the bench line as
    host_numa_node / host_cores so a wrong pin is visible) - re-indexed through ROCR_VISIBLE_DEVICES, then HIP_VISIBLE_DEVICES, when those are plain
    index lists.  -> {'gpu_node': [node of local GPU 0, 1, ...], 'node_cpus': {node: [cpu, ...]}} or None when sysfs does not say."""
    import glob
    try:
        cards = []
        for c in glob.glob('/sys/class/drm/card[0-9]*'):
            if '-' in os.path.basename(c):
                continue
            devp = os.path.join(c, 'device')
            with open(os.path.join(devp, 'vendor')) as f:
                if f.read().strip() != '0x1002':
                    continue
            with open(os.path.join(devp, 'numa_node')) as f:
                node = int(f.read().strip())
            cards.append((os.path.basename(os.path.realpath(devp)), node))
        cards.sort()
        nodes = [n for _, n in cards]
        # the runtime applies ROCR_VISIBLE_DEVICES first (ROCr filters the agents), then HIP_ / CUDA_VISIBLE_DEVICES index into what is left
        for var in ('ROCR_VISIBLE_DEVICES', 'HIP_VISIBLE_DEVICES' if 'HIP_VISIBLE_DEVICES' in os.environ else 'CUDA_VISIBLE_DEVICES'):
            v = os.environ.get(var)
            if v:
                if not all(t.strip().isdigit() for t in v.split(',')) or any(int(t) >= len(nodes) for t in v.split(',')):
                    return None          # UUID lists / out-of-range indices: no guess - the caller falls back to a contiguous slice
                nodes = [nodes[int(t)] for t in v.split(',')]
        if not nodes or any(n < 0 for n in nodes):
            return None
        node_cpus = {}
        for n in set(nodes):
            with open(f'/sys/devices/system/node/node{n}/cpulist') as f:
                cpus = []
                for part in f.read().strip().split(','):
                    lo, _, hi = part.partition('-')
                    cpus += list(range(int(lo), int(hi or lo) + 1))
            node_cpus[n] = cpus
        return {'gpu_node': nodes, 'node_cpus': node_cpus}
    except (OSError, ValueError, IndexError):
        return None


def host_budget(world, local_world, local_rank, cores_avail, allowed, host_cores, no_graphs, workload, topo=None):
    """Per-rank host budget, decided before anything touches the GPU.  `--host-cores K` (the 1-GPU proxy of a rank on a shared node) or,
    when several ranks were launched without it, this rank's slice of the node: usable cores / ranks on the node, if that is <= 8
    (8 ranks x 8 pool threads on the 16 usable cores of a box otherwise).  The K cores come from the NUMA node of the rank's GPU when `topo`
    (gpu_numa_topology) knows it and that node has K usable cores for each of its ranks; else a contiguous slice of the affinity mask.
    -> {'k': cores, 'cores': the CPU ids to pin to (None = leave the affinity alone), 'graphs': switch HIP graphs on (k <= 4: one launch per
    sub-batch instead of ~350), 'auto': decided here, 'numa_node': the node the cores were taken from (None: contiguous slice)}"""
    auto = False
    if world > 1 and host_cores is None:
        k_auto = max(1, cores_avail // max(1, local_world))
        if k_auto <= 8:
            host_cores, auto = k_auto, True
    if host_cores is None:
        return {'k': None, 'cores': None, 'graphs': False, 'auto': False, 'numa_node': None}
    if host_cores < 1:
        raise SystemExit('--host-cores must be >= 1')
    k = min(host_cores, len(allowed))
    mine, node = None, None
    if topo and local_rank < len(topo['gpu_node']) and local_world <= len(topo['gpu_node']):
        node = topo['gpu_node'][local_rank]
        peers = [r for r in range(local_world) if topo['gpu_node'][r] == node]           # the ranks whose GPUs hang off the same node
        near = [c for c in topo['node_cpus'].get(node, []) if c in set(allowed)]
        if len(near) >= k * len(peers):
            j = peers.index(local_rank)
            mine = near[j * k:(j + 1) * k]
        else:
            node = None
    if mine is None:
        mine = [allowed[(local_rank * k + i) % len(allowed)] for i in range(k)]      # (wraps: never a short slice at the tail)
    return {'k': k, 'cores': mine, 'graphs': k <= 4 and not no_graphs and workload in ('fused', 'tiles', 'rowref'), 'auto': auto, 'numa_node': node}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=100, help='timed steps (default 100: a >= 4 s timed region; windows of 20 steps are also reported)')
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--cpu-budget-s', type=float, default=25.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--host-threads', type=int, default=None, help='post-processing pool threads per pipeline (default 8; max(1, K - 1) under --host-cores K)')
    ap.add_argument('--host-cores', type=int, default=None,
                    help='per-rank host budget: pin this process (rank r) to K of the cores it may use, cores [r*K, (r+1)*K), BEFORE any GPU call - '
                         'the 1-GPU proxy for an 8-rank node where every rank only has usable_cores/8.  K <= 4 switches HIP graphs on (one '
                         'launch per sub-batch instead of ~350) unless --no-graphs')
    ap.add_argument('--no-graphs', action='store_true', help='launch kernel by kernel (HIP graphs are the default for the fused / tiles / rowref workloads since round 5)')
    ap.add_argument('--workload', choices=['fused', 'tiles', 'rowref', 'lidar'], default='fused',
                    help="fused = BASELINE configs[2], the headline (on-GPU LAS->BEV raster + config 2, batch 16); tiles = configs[1] "
                         "(pre-rasterised, batch 8); rowref = configs[3] (Proj28_GFC-T3_RowRef head, pre-rasterised, batch 8); lidar = "
                         "configs[4] (sparse-conv LiDAR encoder path, batch 8 point clouds, parity unpinned)")
    ap.add_argument('--streams', type=int, default=None,
                    help='split every batch over this many HIP streams (fills launch tails); default per workload: fused 2, tiles 2, '
                         'rowref 4, lidar 1 (its data-dependent launch sizes need host round trips, which serialise sub-batches)')
    ap.add_argument('--points', choices=['resident', 'host', 'las'], default='resident',
                    help="fused workload only - where a step's point clouds come from.  resident (default, the contract's `value`): already in HBM.  "
                         "host: 16 x 4,194,304 x 16 B of [x,y,z,intensity] f32 in PINNED HOST memory, uploaded every step on a copy stream into one of "
                         "two device buffers while the previous step computes (the reference's read_las -> to_cuda, laserlane_proposals.py:618-636, "
                         "runner.py:125-152).  las: the pinned host memory holds raw LAS point records (format 0, 20 B per point); they are uploaded "
                         "the same way and decoded on the GPU (lm_las_decode_points) in front of the raster.  Both report the achieved H2D rate and "
                         "the fraction of a resident-points window measured in the same run")
    ap.add_argument('--no-second-line', action='store_true',
                    help='skip the second line (fused workload, 1 GPU: the same command once more in a child process under LANEMAP_WINO_SPLIT=1 - '
                         'the Winograd products as fp16 x 2 split terms, fp32-accurate, NOT bit-identical to the exact path; reported as `second_line`, '
                         'never as `value`)')
    ap.add_argument('--conv-detail', action='store_true', help='per-shape table of the MFMA launches on stderr')
    ap.add_argument('--no-stream-check', action='store_true', help='skip the bitwise multi-stream == single-stream check')
    ap.add_argument('--graphs', action='store_true', help='replay the device part of every sub-batch as one HIP graph (TilePipeline use_graph); the default since '
                    'round 5 for the workloads that can be captured (fused, tiles, rowref): +1.4 %% on the headline in four interleaved 100-step runs, '
                    'profiles/r5_graphs_ab.txt')
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        ndev = torch.cuda.device_count()        # (counting devices does not initialise the GPU)
        if ndev < args.gpus and 'LANEMAP_BENCH_DEVICE' not in os.environ:
            raise SystemExit(f'--gpus {args.gpus} but only {ndev} GPU(s) are visible')
        self_launch(args)                       # never returns
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}: launch with `python bench.py --gpus N` (spawns the ranks itself) '
                         f'or `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`')
    # ---- per-rank host budget (nothing has touched the GPU yet: no HIP call, no torch.cuda.is_available)
    cores_avail = usable_cores()
    affinity0 = set(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else None     # (restored for the cpu_baseline leg)
    hb = host_budget(world, int(os.environ.get('LOCAL_WORLD_SIZE', world)), local_rank, cores_avail, sorted(os.sched_getaffinity(0)),
                     args.host_cores, args.no_graphs, args.workload, topo=gpu_numa_topology() if world > 1 else None)
    host_cores_auto = hb['auto']
    if hb['cores'] is not None:
        args.host_cores = hb['k']
        os.sched_setaffinity(0, set(hb['cores']))
        torch.set_num_threads(max(1, hb['k']))
        if hb['graphs']:
            args.graphs = True
    # round 5: graph replay is the default wherever the device part can be captured (the LiDAR path sizes launches on the host): 389.3-390.3
    # against 382.0-385.3 tiles/s in four interleaved pairs of 100-step runs on one box (profiles/r5_graphs_ab.txt); `--no-graphs` = eager
    if not args.no_graphs and args.workload in ('fused', 'tiles', 'rowref'):
        args.graphs = True
    host_cores_per_rank = len(os.sched_getaffinity(0)) if args.host_cores is not None else min(cores_avail, len(os.sched_getaffinity(0)))
    if args.host_threads is None:
        args.host_threads = 8 if args.host_cores is None else max(1, host_cores_per_rank - 1)
    import torch.distributed as dist
    # test hooks (tests/test_gpu_9_bench.py exercises the N>1 code path on a 1-GPU box): every rank on one device, gloo backend
    dev_index = int(os.environ.get('LANEMAP_BENCH_DEVICE', local_rank))
    backend = os.environ.get('LANEMAP_BENCH_BACKEND', 'nccl')
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        torch.cuda.set_device(dev_index)
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', dev_index))
        else:
            dist.init_process_group(backend=backend)
    dev = torch.device('cuda', dev_index)
    torch.cuda.set_device(dev)

    from lanemapping_amd import synth, ops, shard, trace
    from lanemapping_amd._lib import lib
    from lanemapping_amd.boundary import build_net_from_config
    from lanemapping_amd.pipeline import TilePipeline
    lib()                                                    # raises if the HIP library is missing
    cfg_name = {'lidar': 'Proj_polyline_lidarconv_vit_vertex_2', 'rowref': 'Proj28_GFC-T3_RowRef_82_73_laser'}.get(
        args.workload, 'Proj_polyline_fpn_vit_vertex_2')
    net = build_net_from_config(cfg_name, device='cpu')
    synth.fill_module_(net, 2021)
    net = net.to(dev)
    # weak scaling: every rank owns its own BATCH tiles per step (seeds differ per rank)
    batch = 16 if args.workload == 'fused' else BATCH
    if args.workload == 'lidar':    # the config-3 cloud in the ego frame, cropped by the voxeliser, resident in HBM
        clouds = [torch.from_numpy(synth.lidar_points(2021 + rank * 4 + i, N_PTS)).to(dev) for i in range(4)]
        tiles = [clouds[i % 4] for i in range(batch)]
    elif args.workload in ('tiles', 'rowref'):
        tiles = torch.from_numpy(synth.bev_batch([2021 + rank * batch + i for i in range(batch)], 1152)).to(dev)
    else:       # 4,194,304 LAS-shaped points per tile, resident in HBM (4 distinct clouds, repeated)
        clouds = [torch.from_numpy(synth.las_points(2021 + rank * 4 + i, N_PTS)) for i in range(4)]
        points = torch.cat([clouds[i % 4] for i in range(batch)]).to(dev)
        offs = [i * N_PTS for i in range(batch + 1)]
        rpar = [ops.make_raster_params(local_min_ele=-0.5, ele_reso=0.02)] * batch
        # the rasteriser hands the tile over as u8 HWC (what a BEV tile IS: the reference's PNG; f32 = u8 / 255 is applied inside the
        # stem kernel, same bits) - a quarter of the bytes of the f32 planar tensor on both sides
        tiles = torch.empty((batch, 1152, 1152, 3), device=dev, dtype=torch.uint8)
    if args.points != 'resident' and args.workload != 'fused':
        raise SystemExit('--points host / las feeds the rasteriser: it needs --workload fused')
    feed = None             # input-inclusive variants: pinned host source, two device buffers, a copy stream
    if args.points != 'resident':
        from lanemapping_amd import las_io
        feed = {'stream': torch.cuda.Stream(device=dev), 'up_done': [None, None], 'read_done': [None, None], 'pairs': [], 'on': True}
        if args.points == 'host':
            feed['host'] = torch.cat([clouds[i % 4] for i in range(batch)]).pin_memory()
            feed['dev'] = [points, torch.empty_like(points)]
        else:
            # LAS point-data records, format 0 (ASPRS LAS 1.2: X Y Z int32, intensity u16, 6 more bytes), scale 1 mm, offset 0
            las_scale, rl = 1e-3, 20
            recs = [synth.las_point_records(c.numpy(), las_scale) for c in clouds]
            feed['host'] = torch.from_numpy(np.concatenate([recs[i % 4] for i in range(batch)])).pin_memory()
            feed['dev'] = [torch.empty(feed['host'].shape, device=dev, dtype=torch.uint8) for _ in range(2)]
            feed['decode'] = lambda rec_dev: [las_io.decode_points(rec_dev[i * N_PTS * rl:(i + 1) * N_PTS * rl], rl, N_PTS, [las_scale] * 3, [0.0] * 3,
                                                                    None, False, out=points[i * N_PTS:(i + 1) * N_PTS]) for i in range(batch)]
        feed['bytes'] = feed['host'].numel() * feed['host'].element_size()

        def upload(slot):
            """host -> device buffer `slot` on the copy stream, behind the last raster / decode that read that buffer"""
            with torch.cuda.stream(feed['stream']):
                if feed['read_done'][slot] is not None:
                    feed['stream'].wait_event(feed['read_done'][slot])
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                feed['dev'][slot].copy_(feed['host'], non_blocking=True)
                b.record()
                feed['up_done'][slot] = b
                if feed['on']:
                    feed['pairs'].append((a, b))
        feed['upload'] = upload
    pipe = TilePipeline(net, host_threads=args.host_threads, use_graph=args.graphs)
    # whole-batch reference / instrumented passes launch kernel by kernel (the roofline hook brackets every launch with events)
    pipe_eager = pipe if not args.graphs else TilePipeline(net, host_threads=args.host_threads, use_graph=False)
    # default streams per workload (measured with the round-4 kernels, 10-step runs on one box): fused (batch 16) 2 streams 362-363 tiles/s,
    # 4 streams 355-357, 3: 344, 1: 341; tiles (batch 8) 347-348 for 2 / 3 / 4; rowref 344 with 4, 331 with 2
    nstream = max(1, args.streams if args.streams is not None else {'lidar': 1, 'fused': 2, 'tiles': 2}.get(args.workload, 4))
    nstream = min(nstream, batch)
    extra_streams = [torch.cuda.Stream(device=dev) for _ in range(nstream - 1)]
    extra_pipes = [TilePipeline(net, host_threads=args.host_threads, use_graph=args.graphs) for _ in range(nstream - 1)]
    rast = {'pairs': [], 'on': False}
    bounds = [round(i * batch / nstream) for i in range(nstream + 1)]      # every tile of the batch goes to exactly one stream

    # ---- roofline instrumentation: HIP events (on the launch stream = torch's current stream) around every MFMA conv/GEMM launch
    # and every Winograd input transform ----
    prof = {'on': False, 'pairs': [], 'launches': 0}

    def hook(kind, flops, launch, executed=None):
        # flops: algorithmic (direct-convolution) count; executed: what the matrix cores really do (Winograd launches: 16/36)
        if prof['on']:
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            launch()
            b.record()
            prof['pairs'].append((a, b, kind, flops, flops if executed is None else executed))
            prof['launches'] += 1
        else:
            launch()
    ops.set_conv_hook(hook)

    # Stream choreography: sub-batches are independent, so streams never wait for each other in the tiles workload.
    # Fused workload: the raster (main stream) fills one of two alternating tile buffers; a sub-stream waits for the
    # raster's event, and the raster waits for the sub-streams' readers of the same buffer from two steps earlier.
    tile_bufs = [tiles, torch.empty_like(tiles)] if args.workload == 'fused' else [tiles, tiles]
    done = [[None, None] for _ in extra_streams]
    state = {'i': 0}

    def step():
        par = state['i'] & 1
        state['i'] += 1
        cur = tile_bufs[par]
        main = torch.cuda.current_stream()
        ready = None
        if args.workload == 'fused':
            for d in done:
                if d[par] is not None:
                    main.wait_event(d[par])
            src = points
            if feed is not None and feed['on']:
                # this step's clouds were uploaded while the previous step ran (the first one by the priming call); the NEXT step's go out
                # now, into the other buffer, behind the previous step's reader of it
                if feed['up_done'][par] is None:
                    feed['upload'](par)
                main.wait_event(feed['up_done'][par])
                feed['up_done'][par] = None
                feed['upload'](par ^ 1)
                src = feed['dev'][par]
                if args.points == 'las':
                    with trace.stage('las_decode'):
                        feed['decode'](src)
                    src = points
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            with trace.stage('raster'):
                ops.bev_raster_batch(src, offs, rpar, out_u8=cur, u8_only=True)
            b.record()
            ready = b
            if feed is not None and feed['on']:
                if args.points == 'las':            # the uploaded records were consumed by the decode; `points` by the raster (same stream)
                    ev = torch.cuda.Event()
                    ev.record()
                    feed['read_done'][par] = ev
                else:
                    feed['read_done'][par] = b
            if rast['on']:
                rast['pairs'].append((a, b))
        if nstream == 1:
            futs = pipe.submit(cur)
        else:       # independent sub-batches on separate streams: kernels of one fill the partial last wave of the other
            futs = []
            for si in range(nstream):
                sub = cur[bounds[si]:bounds[si + 1]]
                if si == 0:
                    futs += pipe.submit(sub)
                else:
                    es = extra_streams[si - 1]
                    if ready is not None:
                        es.wait_event(ready)
                    with torch.cuda.stream(es):
                        futs += extra_pipes[si - 1].submit(sub)
                        if args.workload == 'fused':
                            ev = torch.cuda.Event()
                            ev.record()
                            done[si - 1][par] = ev
        res = [f.result() for f in futs]
        gather(res)
        return res

    comm = torch.cuda.Stream(device=dev) if world > 1 else None
    inflight = []

    def gather(res):
        """ONE all-gather of the fixed-shape per-tile byte blocks per batch, on its own stream: neither the staging copy nor the
        collective queues behind (or in front of) the compute streams, and the host does not wait for them."""
        if world > 1 and res:
            with torch.cuda.stream(comm):
                block = shard.pack_tile_results([r[0] for r in res], [r[1] for r in res], batch, dev, pinned=True)
                inflight.append((block, shard.all_gather_results(block), res))     # ONE collective per batch
            del inflight[:-4]

    def drain():
        futs = pipe.flush()
        for si, ep in enumerate(extra_pipes):
            with torch.cuda.stream(extra_streams[si]):
                futs += ep.flush()
        res = [f.result() for f in futs]
        gather(res)
        return res

    def single_stream(src):
        """One whole batch on the main stream through the first pipeline -> per-tile results in tile order."""
        for f in pipe_eager.submit(src):
            f.result()
        return [f.result() for f in pipe_eager.flush()]

    # set-up, not measurement: one priming batch loads every kernel's code object, packs the weights into their kernel
    # layouts (PackedModule) and grows the allocator / pinned-buffer pools, so that even `--warmup 0` times steady state
    step()
    drain()
    torch.cuda.synchronize()
    state['i'] = 0
    for _ in range(args.warmup):
        step()
    drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    rast['on'] = True
    prof['on'] = nstream == 1 and not args.graphs      # with >1 streams kernels overlap (and a graph replay bypasses the hook): the roofline is measured in its own pass below
    trace.push('timed_steps')
    marks = []                      # host clock after every 20th step (no synchronisation added: a step returns when the PREVIOUS batch's tiles are done)
    t0 = time.perf_counter()
    for i_ in range(args.steps):
        step()
        if (i_ + 1) % 20 == 0:
            marks.append(time.perf_counter())
    last = drain()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    trace.pop()
    # tiles/s of every complete 20-step window inside the timed region (this rank): separates a 1 % change from box noise
    win = [20 * batch * world / (b_ - a_) for a_, b_ in zip([t0] + marks[:-1], marks)]
    windows = ({'steps_per_window': 20, 'n': len(win), 'min': min(win), 'median': float(np.median(win)), 'max': max(win)} if len(win) >= 2 else None)
    prof['on'] = False
    rast['on'] = False
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- N > 1: what the last all-gather delivered.  Every rank holds the rank-major concatenation: all world x batch slots must be valid
    # tiles, this rank's slice must be the block it sent bit for bit and must unpack to the results of its last batch; the verdict
    # of all ranks is combined so that rank 0's line speaks for the job
    rank_devices = [dev_index]
    if world > 1:
        t = torch.zeros(world, device=dev, dtype=torch.int32)
        t[rank] = dev_index
        dist.all_reduce(t)
        rank_devices = [int(v) for v in t.cpu()]
    gather_check = 'n/a (1 rank)'
    if world > 1 and not inflight:
        gather_check = 'n/a (no batch was gathered: --steps 0)'
    elif world > 1:
        ok = 0
        if inflight:
            comm.synchronize()
            block, gathered, res_l = inflight[-1]
            slots = shard.unpack_gathered(gathered, include_padding=True)
            mine = slots[rank * batch:(rank + 1) * batch]
            ok = int(tuple(gathered.shape) == (world * batch, shard.TILE_BYTES) and all(s_ is not None for s_ in slots)
                     and torch.equal(gathered[rank * batch:(rank + 1) * batch], block) and len(res_l) == batch
                     and all(np.array_equal(m[0], np.asarray(r_[0], dtype=np.float64)) and
                             np.array_equal(m[1], np.asarray(r_[1], dtype=np.int32).reshape(-1, 2)) for m, r_ in zip(mine, res_l)))
        t = torch.tensor([ok], device=dev, dtype=torch.int32)
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) != 1:
            raise SystemExit(f'gather check FAILED on rank {rank}: the all-gathered block does not hold world x batch valid tiles equal to the per-rank results')
        gather_check = (f'last all-gather: {world * batch} valid tiles in one [{world * batch}, {shard.TILE_BYTES}] byte block; on every rank its own slice '
                        f'is bitwise the block it sent and unpacks to the lanes / endpoints of its last batch')

    # ---- fused workload: the rasteriser's tiles of the last timed step against the scalar C oracle (oracle/raster_ref.c) - the first
    # and the last tile of the 16-tile launch (a wrong record slot at a high chunk index, or in the last BatchArgs entry, shows here).
    # Checker leg, outside the clock.
    raster_check = 'n/a'
    if args.workload == 'fused' and args.steps > 0 and not args.no_stream_check:
        from oracle import raster_ref
        lb = tile_bufs[(state['i'] - 1) & 1]
        rp = raster_ref.params(local_min_ele=-0.5, ele_reso=0.02)
        for t_ in (0, batch - 1):
            # (--points las: the raster saw the GPU-decoded records - millimetre-quantised coordinates - so the oracle gets those)
            src_pts = points[t_ * N_PTS:(t_ + 1) * N_PTS].cpu().numpy() if args.points == 'las' else clouds[t_ % 4].numpy()
            want = raster_ref.raster(src_pts, rp, 1152, 1152)
            got = lb[t_].cpu().numpy()
            if not np.array_equal(got, want):
                raise SystemExit(f'raster check FAILED: tile {t_} of the last timed step differs from oracle/raster_ref.c in '
                                 f'{int((got != want).any(axis=2).sum())} pixels')
        raster_check = (f'tiles 0 and {batch - 1} of the last timed {batch} x {N_PTS}-point launch equal oracle/raster_ref.c bit for bit '
                        f'(u8 HWC, {1152 * 1152} pixels each)')

    # ---- the product path the clock just timed splits each batch over `nstream` streams / pipelines: its outputs for the last step
    # must equal a single-stream run on the same tiles BITWISE (a cross-stream race on workspaces or packed weights would show here)
    last_buf = tile_bufs[(state['i'] - 1) & 1] if args.steps > 0 else tiles
    stream_check = 'skipped'
    if nstream > 1 and not args.no_stream_check and last:
        ref = single_stream(last_buf)
        if len(ref) != len(last):
            raise SystemExit(f'stream check: {len(last)} results from the {nstream}-stream step, {len(ref)} from the single-stream run')
        for t_, ((la, ea), (lb, eb)) in enumerate(zip(last, ref)):
            if not (np.array_equal(np.asarray(la), np.asarray(lb)) and np.array_equal(np.asarray(ea), np.asarray(eb))):
                raise SystemExit(f'stream check FAILED: tile {t_} of the last timed step differs between the {nstream}-stream product path '
                                 f'and a single-stream run (lanes equal: {np.array_equal(np.asarray(la), np.asarray(lb))})')
        stream_check = f'lanes and endpoints of the last timed step ({len(last)} tiles, {nstream} streams) bitwise equal to a single-stream run'

    roof_steps = args.steps
    roof_scope = 'HIP events around every launch inside the timed region (single stream)'
    if nstream > 1 or args.graphs:
        # per-launch durations are only meaningful when launches do not share the GPU: instrumented single-stream steps
        roof_steps = 2
        roof_scope = (f'{roof_steps} instrumented single-stream steps right after the timed region (in the timed region the batch is '
                      f'split over {nstream} streams whose kernels overlap, so per-launch durations are not additive)')
        # one untimed single-stream step first: whole-batch tensors have shapes the 4-stream steps never allocated, and a
        # hipMalloc inside an event bracket (it synchronises the device) would be charged to that launch
        single_stream(tiles if args.workload != 'fused' else last_buf)
        torch.cuda.synchronize()
        prof['on'] = True
        if args.workload == 'fused':
            rast['pairs'] = []          # in the timed region the raster shares the GPU with the other streams' kernels
        for _ in range(roof_steps):
            if args.workload == 'fused':
                # the GPU is idle at this point (the previous batch's results were waited for): an untimed launch first, so that the
                # timed one is already queued behind it when its start event fires - the bracket then holds the two kernels'
                # execution, not the host's argument marshalling for 16 tiles
                ops.bev_raster_batch(points, offs, rpar, out_u8=tiles, u8_only=True)
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record()
                ops.bev_raster_batch(points, offs, rpar, out_u8=tiles, u8_only=True)
                b.record()
                rast['pairs'].append((a, b))
            for f in pipe_eager.submit(tiles):
                f.result()
        for f in pipe_eager.flush():
            f.result()
        torch.cuda.synchronize()
        prof['on'] = False

    # ---- input-inclusive variants: the upload's own rate, and a resident-points window of the same command right behind it
    feed_report = None
    if feed is not None:
        up_ms = [a.elapsed_time(b) for a, b in feed['pairs']]
        feed['on'] = False
        torch.cuda.synchronize()
        n_res = min(40, max(10, args.steps))
        for _ in range(3):
            step()
        drain()
        torch.cuda.synchronize()
        t_r = time.perf_counter()
        for _ in range(n_res):
            step()
        drain()
        torch.cuda.synchronize()
        res_rate = batch * n_res / (time.perf_counter() - t_r)
        my_rate = batch * args.steps / dt
        feed_report = {'source': {'host': 'pinned host memory, [x,y,z,intensity] f32 (16 B per point)',
                                  'las': 'pinned host memory, LAS 1.2 point format 0 records (20 B per point), decoded on the GPU by lm_las_decode_points'}[args.points],
                       'bytes_per_step': feed['bytes'], 'uploads_timed': len(up_ms),
                       'h2d_ms_per_step': float(np.mean(up_ms)) if up_ms else None,
                       'h2d_GBps': feed['bytes'] / (float(np.mean(up_ms)) * 1e-3) / 1e9 if up_ms else None,      # while the compute streams run
                       'h2d_GBps_needed_at_this_rate': feed['bytes'] * my_rate / batch / 1e9,
                       'double_buffered': True, 'copy_stream': True,
                       'resident_window': {'steps': n_res, 'tiles_per_s_this_rank': res_rate},
                       'fraction_of_resident': my_rate / res_rate}

    # ---- aggregate per kernel class.  kind strings: 'wino44 ...', 'conv ...', 'gemm ...', 'spconv ...'
    def kclass(kind):
        k = kind.split(' ', 1)[0]
        return {'wino44': 'wino44_kernel', 'wino44split': 'wino44_kernel<split: fp16 x 2 terms, 3 products>'}.get(k, 'conv_mfma_kernel')
    # every MFMA launch of the headline path is exact fp32; under LANEMAP_WINO_SPLIT=1 (second line) the Winograd launches run fp16 MFMAs
    peak_of = lambda c: MFMA_F16_PEAK_TFLOPS if 'split' in c else MFMA_F32_PEAK_TFLOPS
    cls = {}
    per_kind = {}
    for a, b, kind, fl, ex in prof['pairs']:
        ms = a.elapsed_time(b)
        e = cls.setdefault(kclass(kind), [0, 0.0, 0.0, 0.0])
        e[0] += 1; e[1] += ms; e[2] += fl; e[3] += ex
        e = per_kind.setdefault(kind, [0, 0.0, 0.0, 0.0])
        e[0] += 1; e[1] += ms; e[2] += fl; e[3] += ex
    if args.conv_detail and rank == 0:
        for kind, (n, ms, fl, ex) in sorted(per_kind.items(), key=lambda kv: -kv[1][1]):
            print(f'# {kind:52s} x{n // max(roof_steps, 1):3d}/step {ms / roof_steps:8.3f} ms/step {ex / max(ms, 1e-9) / 1e9:7.1f} TFLOP/s executed '
                  f'({fl / max(ms, 1e-9) / 1e9:7.1f} direct-equivalent)', file=sys.stderr)
    conv_ms = sum(e[1] for e in cls.values())
    alg = sum(e[2] for e in cls.values())
    exe = sum(e[3] for e in cls.values())
    executed_tflops = exe / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0           # what the matrix cores really issue
    alg_tflops = alg / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0                # direct-convolution equivalent
    rs = max(roof_steps, 1)
    per_class = {k: {'launches_per_step': e[0] / rs, 'ms_per_step': e[1] / rs, 'avg_launch_ms': e[1] / max(e[0], 1),
                     'executed_gflop_per_step': e[3] / rs / 1e9,
                     'executed_tflops': e[3] / (e[1] * 1e-3) / 1e12 if e[1] > 0 else 0.0,
                     'peak': peak_of(k), 'frac': e[3] / (e[1] * 1e-3) / 1e12 / peak_of(k) if e[1] > 0 else 0.0}
                 for k, e in cls.items()}
    dominant = max(per_class, key=lambda k: per_class[k]['ms_per_step']) if per_class else None
    # HBM bytes of the same kernels from the PMC counters (collected offline in their own rocprofv3 --pmc passes, FETCH_SIZE x 2 per the
    # gfx950 note of MI355X_MICROARCH.md + WRITE_SIZE; tools/pmc_traffic.sh writes the file, profiles/ holds the raw summaries)
    pmc = {}
    try:
        with open(PMC_FILE) as f:
            pmc = json.load(f).get(args.workload, {})
    except (OSError, ValueError):
        pass
    mfma_traffic = pmc.get('mfma_bytes_per_step')
    # the counters were collected on other kernel sources than the ones that ran (None: no counters were collected for this workload)
    traffic_stale = ((pmc['kernel_sha16'] != kernel_sha16()) if 'kernel_sha16' in pmc else (pmc.get('csrc_sha16') != csrc_sha16())) if pmc else None
    if os.environ.get('LANEMAP_WINO_F44', '1') == '0':
        mfma_traffic, traffic_stale = None, None        # the counters were collected on the default kernels (F(4x4)), not on this run's
    n_lines = int(np.mean([(np.count_nonzero(l[:, :, 0] > 0, axis=1) >= 2).sum() for l, _ in last])) if last else 0
    what = {'tiles': 'pre-rasterised tile', 'fused': 'LAS points', 'lidar': 'LiDAR point cloud', 'rowref': 'pre-rasterised tile'}[args.workload]
    workload = {
        'tiles': 'configs/Proj_polyline_fpn_vit_vertex_2.py inference, batch=8 per GPU, pre-rasterised synthetic 1152x1152 BEV tiles '
                 'resident in HBM, seeded random weights',
        'rowref': 'configs/Proj28_GFC-T3_RowRef_82_73_laser.py inference (RowSharNotReducRef head), batch=8 per GPU, pre-rasterised '
                  'synthetic 1152x1152 BEV tiles resident in HBM, seeded random weights',
        'lidar': 'configs/Proj_polyline_lidarconv_vit_vertex_2.py inference (sparse-conv LiDAR encoder, parity unpinned), batch=8 '
                 'point clouds of 4,194,304 points per GPU resident in HBM, seeded random weights',
        'fused': 'on-GPU LAS->BEV raster (4,194,304 points/tile resident in HBM) + configs/Proj_polyline_fpn_vit_vertex_2.py, '
                 'batch=16 per GPU, seeded random weights'}[args.workload]
    if args.points == 'host':
        what = 'LAS points in pinned host memory'
        workload = workload.replace('resident in HBM', 'in PINNED HOST memory, uploaded every step (double-buffered, copy stream)')
    elif args.points == 'las':
        what = 'LAS point records in pinned host memory'
        workload = workload.replace('resident in HBM', 'as LAS format-0 records in PINNED HOST memory, uploaded every step and decoded on the GPU')
    result = {
        'metric': f'BEV tiles/sec end-to-end ({what} -> polylines)',
        'value': world * batch * args.steps / dt,
        'unit': 'tiles/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32' if not SPLIT_LINE else 'f16x2-split Winograd products (3 per fp32 product), f32 accumulate; everything else f32', 'data': 'synthetic',
        'config': {'workload': workload,
                   'tiles_per_step_per_gpu': batch, 'lines_per_tile': n_lines, 'host_threads': args.host_threads, 'streams': nstream, 'hip_graphs': bool(args.graphs and args.workload in ('fused', 'tiles', 'rowref')),      # (the LiDAR path sizes launches on the host: no capture)
                  
                   'points': args.points if args.workload == 'fused' else None, 'point_feed': feed_report,
                   'windows_tiles_per_s': windows,
                   'stream_check': stream_check, 'gather_check': gather_check, 'raster_check': raster_check,
                   'host_cores_per_rank': host_cores_per_rank, 'host_cores_pinned': args.host_cores is not None,
                   'host_cores_auto': host_cores_auto,       # N > 1 without --host-cores: usable cores / ranks on the node
                   'host_numa_node': hb.get('numa_node'),
                   # what the collective library saw: the proof that an N-GPU line really ran N RCCL ranks
                   # (rccl_ranks counts RCCL ranks only: 0 when the test hook put the ranks on gloo; rank_devices = HIP device index of every rank)
                   'dist_ranks': world, 'dist_backend': dist.get_backend() if world > 1 else None,
                   'rccl_ranks': (world if dist.get_backend() == 'nccl' else 0) if world > 1 else 1, 'rank_devices': rank_devices,
                   'host_cores': sorted(os.sched_getaffinity(0)) if args.host_cores is not None else None,
                   'host_postproc_ms_per_tile': 1e3 * sum(p_.host_seconds for p_ in [pipe] + extra_pipes) /
                   max(1, sum(p_.host_tiles for p_ in [pipe] + extra_pipes))},
        'roofline': {'bound': 'mfma',
                     'kernel': 'every MFMA convolution / GEMM launch of a step (' + ', '.join(sorted(per_class)) + ')',
                     # EXECUTED view: FLOPs the matrix cores really issue (Winograd F(4x4,3x3) launches: 36/144 of the direct count)
                     # / summed HIP-event time of those launches
                     'achieved': executed_tflops, 'peak': MFMA_F32_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                     # time-weighted utilisation: sum_k (executed_k / peak_k) / sum_k t_k  (== achieved / peak when every launch is fp32 MFMA)
                     'frac': (sum(e[3] / peak_of(k) for k, e in cls.items()) / (conv_ms * 1e-3) / 1e12) if conv_ms > 0 else 0.0,
                     'traffic': mfma_traffic, 'traffic_source': pmc.get('source'), 'traffic_stale': traffic_stale,
                     'algorithmic_equiv_tflops': alg_tflops, 'frac_survey_8d': alg_tflops / MFMA_F32_PEAK_TFLOPS,
                     'scope': roof_scope, 'launches_per_step': prof['launches'] / rs,
                     'executed_gflop_per_step': exe / rs / 1e9, 'algorithmic_gflop_per_step': alg / rs / 1e9,
                     'kernel_ms_per_step': conv_ms / rs,
                     'winograd_ms_per_step': sum(v['ms_per_step'] for k, v in per_class.items() if k.startswith('wino')),
                     'dominant_kernel': dominant, 'per_kernel': per_class,
                     'note': 'achieved / frac = executed MFMA FLOPs / launch time / fp32 MFMA peak (always <= 1). algorithmic_equiv_tflops = '
                             'direct-convolution FLOPs (SURVEY 8d: 2 per MAC of the 3x3 sums) / the same time: it exceeds the executed figure '
                             'because Winograd F(4x4,3x3) does 36 multiplies where the direct sum does 144; frac_survey_8d = algorithmic_equiv_tflops / peak (SURVEY 8(d)\'s definition: it can exceed 1 for the same reason)'},
    }
    if args.workload == 'fused' and rast['pairs']:
        rms = sum(a.elapsed_time(b) for a, b in rast['pairs']) / len(rast['pairs'])
        algb = (16.0 * N_PTS + 3 * 1152 * 1152 * 4) * batch
        movedb = (16.0 * N_PTS + 3 * 1152 * 1152) * batch          # what this design has to move: the tile leaves as u8
        result['raster_roofline'] = {'bound': 'hbm', 'kernel': 'raster_partition_kernel + raster_band_kernel',
                                     # frac / achieved: the bytes THIS design has to move (16 B per point + the 3 x H x W u8 tile);
                                     # frac_survey_8d: SURVEY 8(d)'s numerator (16 B per point + a 3 x H x W f32 tile that is never written)
                                     'achieved': movedb / (rms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                     'frac': movedb / (rms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                     'frac_survey_8d': algb / (rms * 1e-3) / 1e9 / HBM_PEAK_GBS, 'survey_8d_bytes_per_step': algb,
                                     'traffic': pmc.get('raster_bytes_per_step'), 'traffic_source': pmc.get('source'), 'traffic_stale': traffic_stale,
                                     'algorithmic_bytes_per_step': movedb, 'ms_per_step': rms,
                                     'scope': roof_scope + ('; each timed raster launch is queued behind an untimed one, so the event bracket holds GPU execution only' if (nstream > 1 or args.graphs) else ''),
                                     'note': 'achieved / frac = the bytes this design must move (16 B per point + the 3 x H x W u8 HWC tile: its only '
                                             'consumer, the stem kernel, applies u8 / 255 itself, bit-identical) / time / peak; frac_survey_8d = '
                                             'SURVEY 8(d) bytes (the same points + a 3 x H x W f32 tile, 4x the tile bytes, never written) / the same '
                                             'time; traffic = what the counters saw'}
    # the CPU path timed on this node's own host cores in the same run, next to the 1 / 2 / 4 / 8-GPU numbers (north_star).  At N > 1 the
    # other ranks must not burn cores meanwhile: an NCCL barrier is a stream synchronise that may spin, so they finish their GPU work in a
    # barrier FIRST and are then parked on a key of the rendezvous store (a blocking socket read) until rank 0 has its number
    store = None
    if world > 1:
        dist.barrier()
        torch.cuda.synchronize()
        store = dist.distributed_c10d._get_default_store()
    if rank == 0 and world == 1 and args.workload == 'fused' and args.points == 'resident' and not args.no_second_line and not SPLIT_LINE:
        result['second_line'] = second_line(args)
    if rank == 0:
        if not args.no_cpu_baseline and affinity0 is not None and args.host_cores is not None:
            os.sched_setaffinity(0, affinity0)          # the CPU path gets the node's cores back, not this rank's slice
        result['cpu_baseline'] = None if args.no_cpu_baseline else cpu_baseline(args.cpu_budget_s)
        if result['cpu_baseline'] is not None and world > 1:
            result['cpu_baseline']['measured_with'] = f'{world - 1} other rank(s) of this job idle on the node (blocked on a store key, not spinning)'
        print(json.dumps(result), flush=True)
        if store is not None:
            store.set('lanemap_cpu_baseline_done', '1')
    elif store is not None:
        store.wait(['lanemap_cpu_baseline_done'], datetime.timedelta(seconds=max(600.0, 40.0 * args.cpu_budget_s)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
