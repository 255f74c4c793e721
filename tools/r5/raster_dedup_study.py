#!/usr/bin/env python3
"""How many of the rasteriser's records are DOMINATED inside the unit that could drop them for free - the chunk one pass-1 workgroup sorts in
LDS (16,384 points, csrc/raster.hip) - and inside larger hypothetical units?  A record is dominated when another point of the same unit
falls on the same pixel with a key (I << 8 | G) at least as large.  CPU / numpy on the bench's own synthetic clouds (synth.las_points:
70 % uniform, 30 % on six 3-pixel-wide stripes, acquisition order = unsorted).  Output: profiles/r5_raster_dedup_study.txt"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from lanemapping_amd import synth  # noqa: E402

H = W = 1152
N = 4194304
RESO = 0.05
rows_all = []
for seed in (0, 1):
    p = synth.las_points(seed, N)
    row = np.floor(p[:, 0] / RESO + 0.5).astype(np.int64)
    col = np.floor(p[:, 1] / RESO + 0.5).astype(np.int64)
    ok = (row >= 0) & (row < H) & (col >= 0) & (col < W)
    pix = (row * W + col)[ok]
    print(f'cloud seed {seed}: {N} points, {ok.sum()} inside the tile, {np.unique(pix).size} distinct pixels of {H * W} '
          f'({ok.sum() / np.unique(pix).size:.2f} points per touched pixel)')
    for chunk in (16384, 65536, 262144, 1048576, N):
        kept = 0
        occ = []
        for s in range(0, pix.size, chunk):
            u, c = np.unique(pix[s:s + chunk], return_counts=True)
            kept += u.size
            occ.append(c)
        occ = np.concatenate(occ)
        hist = np.bincount(np.minimum(occ, 8), minlength=9)[1:]
        frac = 1.0 - kept / pix.size
        # traffic of the two-pass design per tile (profiles/r4_pmc_traffic.txt): 16 N points + 4.2 N records out + 5.0 N records in + 3 H W
        base = 16 * N + 9.2 * N + 3 * H * W
        new = 16 * N + 9.2 * N * (1 - frac) + 3 * H * W
        print(f'  unit = {chunk:8d} points: {100 * frac:5.2f} % of the records dominated -> {100 * (1 - new / base):4.1f} % less HBM traffic '
              f'({base / 1e6:.1f} -> {new / 1e6:.1f} MB per tile); records per (unit, pixel) 1 / 2 / 3 / 4 / 5 / 6 / 7 / 8+: '
              + ' / '.join(str(int(v)) for v in hist))
