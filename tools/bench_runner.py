#!/usr/bin/env python3
"""File-to-file throughput of Runner (config 2): PNG tiles on disk -> polyline JSON files; prints the stage times.
usage: bench_runner.py [n_tiles]"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanemapping_amd import synth  # noqa: E402
from lanemapping_amd.boundary import load_config  # noqa: E402
from lanemapping_amd.runner import Runner  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cfg = load_config(os.path.join(root, 'configs', 'Proj_polyline_fpn_vit_vertex_2.py'))
d = tempfile.mkdtemp()
from PIL import Image  # noqa: E402  (test data only)
for i in range(n):
    Image.fromarray(synth.bev_tile_u8(100 + i % 8, 1152)).save(os.path.join(d, f'{19010000000 + i}_x.png'))
r = Runner(cfg)
synth.fill_module_(r.net, 2021)
out = os.path.join(d, 'out')
r.infer_lane_coordinate_endpoint_semantics(tiles=d, work_dirs=out, batch_size=8, write_lane_vertex=True)        # warm-up (packing, allocator)
torch.cuda.synchronize()
for write in (False, True):
    t0 = time.time()
    res = r.infer_lane_coordinate_endpoint_semantics(tiles=d, work_dirs=out, batch_size=8, write_lane_vertex=write)
    torch.cuda.synchronize()
    dt = time.time() - t0
    print(f'{n} PNG tiles -> polylines{" -> JSON files" if write else ""}: {dt:.2f} s = {n / dt:.1f} tiles/s')
