"""CPU, world_size 2 over gloo: the N>1 path of the bench (static tile shard + ONE all-gather of one fixed-shape byte
block per tile: lanes f64 | endpoints i32 | flags)."""
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
    block = shard.pack_tile_results(lanes, endp, per, torch.device('cpu'))
    assert block.dtype == torch.uint8 and tuple(block.shape) == (per, shard.TILE_BYTES)
    calls = []
    real = dist.all_gather_into_tensor
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    gathered = shard.all_gather_results(block)
    dist.all_gather_into_tensor = real
    assert len(calls) == 1, 'one collective per batch'
    out = shard.unpack_gathered(gathered)
    slots = shard.unpack_gathered(gathered, include_padding=True)
    assert len(slots) == world * per and [s is None for s in slots] == [t >= n_tiles for t in range(world * per)]
    # the rank's own slice of the gathered block is what it sent, bit for bit
    assert torch.equal(gathered[rank * per:(rank + 1) * per], block)
    for (l, e), l0, e0 in zip(out[lo:hi], lanes, endp):
        assert np.array_equal(l, l0) and np.array_equal(e, np.asarray(e0, dtype=np.int32).reshape(-1, 2))
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
    """The block has room for every candidate a tile can have (the clusters of its top-K endpoint pixels), so no rank can fail alone
    in front of the collective; a caller that hands over more is refused loudly instead of being truncated."""
    import pytest
    from lanemapping_amd import decode
    assert shard.MAX_ENDP >= decode.TOPK and shard.TILE_BYTES % 8 == 0
    lanes = [np.full((72, 144, 2), -1.0)]
    ok = shard.pack_tile_results(lanes, [np.arange(2 * decode.TOPK, dtype=np.int32).reshape(-1, 2)], 2, torch.device('cpu'))
    (l, e), pad = shard.unpack_gathered(ok, include_padding=True)
    assert pad is None and np.array_equal(l, lanes[0]) and np.array_equal(e, np.arange(2 * decode.TOPK).reshape(-1, 2))
    with pytest.raises(ValueError):
        shard.pack_tile_results(lanes, [np.zeros((shard.MAX_ENDP + 1, 2), dtype=np.int32)], 1, torch.device('cpu'))
