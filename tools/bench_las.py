#!/usr/bin/env python3
"""Micro-bench of lm_las_decode_points (LAS point records resident in HBM -> [N,4] f32): HBM-bound byte work,
algorithmic bytes = N * (record_len + 16)."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanemapping_amd import las_io  # noqa: E402

dev = torch.device('cuda:0')
N = 1 << 24
for rl in (20, 28, 34):
    rec = torch.randint(0, 255, ((N * rl + 3) // 4 * 4,), dtype=torch.uint8, device=dev)
    for _ in range(3):
        las_io.decode_points(rec, rl, N, [1e-3] * 3, [0.0] * 3, None, False)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        las_io.decode_points(rec, rl, N, [1e-3] * 3, [0.0] * 3, None, False)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10
    print(json.dumps({'record_len': rl, 'points': N, 'ms': ms, 'GBps': N * (rl + 16) / ms / 1e6, 'frac_of_8TBps': N * (rl + 16) / ms / 1e6 / 8000}))
