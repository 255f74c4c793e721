#!/usr/bin/env python3
"""How much of a step is host-side enqueue time?  Times the submit() calls of a 4-stream step (no waiting on results)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanemapping_amd import synth  # noqa: E402
from lanemapping_amd.boundary import build_net_from_config  # noqa: E402
from lanemapping_amd.pipeline import TilePipeline  # noqa: E402

dev = torch.device('cuda:0')
net = build_net_from_config('Proj_polyline_fpn_vit_vertex_2', device='cpu')
synth.fill_module_(net, 2021)
net = net.to(dev)
tiles = torch.from_numpy(synth.bev_batch([2021 + i for i in range(8)], 1152)).to(dev)
NS = 4
streams = [torch.cuda.current_stream()] + [torch.cuda.Stream(device=dev) for _ in range(NS - 1)]
GRAPHS = os.environ.get('GRAPHS', '0') != '0'       # GRAPHS=1: the device part of every sub-batch replayed as one HIP graph
pipes = [TilePipeline(net, use_graph=GRAPHS) for _ in range(NS)]


def step():
    enq = 0.0
    futs = []
    for si in range(NS):
        with torch.cuda.stream(streams[si]), torch.no_grad():
            t0 = time.perf_counter()
            new = pipes[si]._gpu_stage(tiles[2 * si:2 * si + 2])      # pure enqueue: kernels + async D2H + event record
            enq += time.perf_counter() - t0
            if pipes[si]._pending is not None:
                futs += pipes[si]._finish(pipes[si]._pending)         # waits for the previous batch of this stream
            pipes[si]._pending = new
    for f in futs:
        f.result()
    return enq


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
enq = sum(step() for _ in range(10))
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f'graphs={GRAPHS}: step {tot / 10 * 1e3:.2f} ms, of which pure host enqueue (kernel launches, allocations, async copies) {enq / 10 * 1e3:.2f} ms')
