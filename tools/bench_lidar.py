#!/usr/bin/env python3
"""Micro-bench of the config-5 sparse-conv LiDAR path: B samples of N synthetic ego-frame points
resident in HBM -> LidarEncoder (voxelise, SparseEncoder, dense tail) -> ViT -> head -> decode.  Prints stage times and the
per-layer MFMA gather-conv throughput."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanemapping_amd import ops, synth  # noqa: E402
from lanemapping_amd.boundary import build_net_from_config  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
N = int(sys.argv[2]) if len(sys.argv) > 2 else 4194304
dev = torch.device('cuda:0')
net = build_net_from_config('Proj_polyline_lidarconv_vit_vertex_2', device='cpu')
synth.fill_module_(net, 2021)
net = net.to(dev)
base = [torch.from_numpy(synth.lidar_points(2021 + i, N)).to(dev) for i in range(min(B, 4))]
pts = [base[i % len(base)] for i in range(B)]
enc = net.pcencoder


def timed(fn, rep=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(rep):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / rep * 1e3, out


with torch.no_grad():
    t_vox, (feats, coords, ends) = timed(lambda: enc.voxelize(pts, raster_order=True))
    t_sp, dense = timed(lambda: enc.sparse_backbone(feats, coords, B))
    t_tail, _ = timed(lambda: enc.dense_tail(dense))
    t_raw, _ = timed(lambda: net.forward_raw({'points': pts}))
    t_full, _ = timed(lambda: net({'points': pts}), rep=2)
    agg = {}

    def hook(kind, flops, launch):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        launch()
        b.record()
        agg.setdefault(kind, []).append((a, b, flops))
    ops.set_conv_hook(hook)
    enc.sparse_backbone(feats, coords, B)
    torch.cuda.synchronize()
    ops.set_conv_hook(None)
tot_ms = tot_fl = 0.0
for kind, lst in agg.items():
    ms = sum(a.elapsed_time(b) for a, b, _ in lst)
    fl = sum(f for _, _, f in lst)
    tot_ms += ms
    tot_fl += fl
    print(f'# {kind:40s} x{len(lst):2d} {ms:8.3f} ms {fl / ms / 1e9:7.1f} TFLOP/s (padded channels counted)', file=sys.stderr)
print(json.dumps({'samples': B, 'points_per_sample': N, 'voxels': int(coords.shape[0]), 'ms_voxelize': t_vox, 'ms_sparse_encoder': t_sp,
                  'ms_dense_tail': t_tail, 'ms_forward_raw': t_raw, 'ms_full_forward_with_postproc': t_full,
                  'samples_per_s_raw': B / t_raw * 1e3, 'samples_per_s_full': B / t_full * 1e3,
                  'gather_conv_ms': tot_ms, 'gather_conv_TFLOPs': tot_fl / tot_ms / 1e9}))
