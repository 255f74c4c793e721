#!/usr/bin/env python3
"""Micro-bench of the LAS->BEV rasteriser (SURVEY §8d config 3 input: N = 4,194,304 points/tile, 16 B/point).
Prints achieved algorithmic GB/s = (16 N + 3*1152^2*4) bytes / kernel time vs the 8 TB/s HBM peak."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanemapping_amd import ops, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4194304
TILES = 16
dev = torch.device('cuda:0')
if os.environ.get('RASTER_UNIFORM'):      # uniform x, y instead of the config-3 cloud (30 % of the points on 6 lane stripes)
    g = torch.Generator().manual_seed(1)
    one = torch.rand((N, 4), generator=g) * torch.tensor([57.6, 57.6, 1.0, 30000.0]) + torch.tensor([0, 0, 0, 800.0])
    pts = torch.cat([one] * TILES).to(dev)
else:
    pts = torch.cat([torch.from_numpy(synth.las_points(2021 + (i % 4), N)) for i in range(TILES)]).to(dev)
par = [ops.make_raster_params(local_min_ele=-0.5, ele_reso=0.02)] * TILES
offs = [i * N for i in range(TILES + 1)]
MODE = os.environ.get('RASTER_OUT', 'u8')          # u8 = u8 HWC only (what the pipeline uses: the stem takes it); f32 = planar f32 (the reference's tensor)
if MODE == 'u8':
    out = torch.empty((TILES, 1152, 1152, 3), device=dev, dtype=torch.uint8)
    run = lambda: ops.bev_raster_batch(pts, offs, par, out_u8=out, u8_only=True)
else:
    out = torch.empty((TILES, 3, 1152, 1152), device=dev)
    run = lambda: ops.bev_raster_batch(pts, offs, par, out=out)
for i in range(3):
    run()
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
REP = 10
for r in range(REP):
    run()
b.record()
torch.cuda.synchronize()
ms = a.elapsed_time(b) / (REP * TILES)
alg = 16 * N + 3 * 1152 * 1152 * 4
print(json.dumps({'points_per_tile': N, 'tiles_per_launch': TILES, 'output': MODE, 'band_rows': os.environ.get('LM_RASTER_BAND_ROWS', 'auto'),
                  'ms_per_tile': ms, 'algorithmic_GBps': alg / ms / 1e6, 'frac_of_8TBps': alg / ms / 1e6 / 8000, 'tiles_per_s': 1e3 / ms}))
