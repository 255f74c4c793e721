#!/usr/bin/env python3
"""Micro-bench of the config-4 path (Proj28_GFC-T3_RowRef): B pre-rasterised tiles -> FPN -> ViT -> RowSharNotReducRef ->
decode -> per-lane line tracing.  Prints stage times."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanemapping_amd import synth  # noqa: E402
from lanemapping_amd.boundary import build_net_from_config  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda:0')
net = build_net_from_config('Proj28_GFC-T3_RowRef_82_73_laser', device='cpu')
synth.fill_module_(net, 2021)
net = net.to(dev)
x = torch.from_numpy(synth.bev_batch([2021 + i for i in range(B)], 1152)).to(dev)


def timed(fn, rep=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(rep):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / rep * 1e3, out


with torch.no_grad():
    t_enc, enc = timed(lambda: net.pcencoder({'proj': x}))
    t_vit, fea = timed(lambda: net.backbone(enc[0]))
    t_head, out = timed(lambda: net.heads(fea))
    t_raw, _ = timed(lambda: net.forward_raw({'proj': x}))
    t_full, _ = timed(lambda: net({'proj': x}), rep=2)
print(json.dumps({'tiles': B, 'ms_fpn': t_enc, 'ms_vit': t_vit, 'ms_rowref_head': t_head, 'ms_forward_raw': t_raw, 'ms_full_forward': t_full,
                  'tiles_per_s_raw': B / t_raw * 1e3, 'tiles_per_s_full': B / t_full * 1e3}))
