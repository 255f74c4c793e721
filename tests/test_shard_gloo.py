"""CPU, world_size 2 over gloo: the N>1 path of the bench (static tile shard + one all-gather of the
fixed-shape polyline blocks)."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lanemapping_amd import shard


def test_shard_range_covers_all_tiles():
    for n, w in ((1024, 8), (10, 4), (3, 8), (0, 2)):
        got = []
        for r in range(w):
            lo, hi, per = shard.shard_range(n, r, w)
            assert hi - lo <= per and per == (n + w - 1) // w
            got += list(range(lo, hi))
        assert got == list(range(n))


def _worker(rank, world, port, n_tiles, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    lo, hi, per = shard.shard_range(n_tiles, rank, world)
    lanes, endp = [], []
    for t in range(lo, hi):
        l = np.full((72, 144, 2), -1.0)
        l[..., 1] = 0
        l[t % 72, :, 0] = 100.0 + t + 1e-9                # needs float64 to survive the gather
        l[t % 72, :, 1] = 1 + t % 2
        lanes.append(l)
        endp.append(np.array([[t, t + 1]] * (t % 3)))
    blocks = shard.pack_tile_results(lanes, endp, per, torch.device('cpu'))
    out = shard.unpack_gathered(*shard.all_gather_results(*blocks))
    q.put((rank, [(float(l[:, :, 0].max()), len(e)) for l, e in out]))
    dist.barrier()
    dist.destroy_process_group()


def test_all_gather_world2_gloo():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    n_tiles = 5                                   # ragged: rank 0 owns 3 tiles, rank 1 owns 2 + 1 padding slot
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_tiles, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [(100.0 + t + 1e-9, t % 3) for t in range(n_tiles)]     # bit-exact f64 (rank 0 writes byte-identical files)
    assert res[0] == want and res[1] == want      # every rank holds all tiles, global order, padding dropped


def test_pack_refuses_to_drop_endpoints():
    import pytest
    lanes = [np.full((72, 144, 2), -1.0)]
    with pytest.raises(ValueError):
        shard.pack_tile_results(lanes, [np.zeros((shard.MAX_ENDP + 1, 2), dtype=np.int32)], 1, torch.device('cpu'))
