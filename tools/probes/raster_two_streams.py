import os, sys, json, torch
sys.path.insert(0, '/root/repo')
from lanemapping_amd import ops, synth
N = 4194304; TILES = 16
dev = torch.device('cuda:0')
pts = torch.cat([torch.from_numpy(synth.las_points(2021 + (i % 4), N)) for i in range(TILES)]).to(dev)
par = [ops.make_raster_params(local_min_ele=-0.5, ele_reso=0.02)] * TILES
out = torch.empty((TILES, 3, 1152, 1152), device=dev)
s2 = torch.cuda.Stream()
def one():
    ops.bev_raster_batch(pts, [i * N for i in range(TILES + 1)], par, out=out)
def two(G):
    ev = torch.cuda.Event(); ev.record()
    n = TILES // G
    for g in range(G):
        st = torch.cuda.current_stream() if g % 2 == 0 else s2
        if g % 2: s2.wait_event(ev)
        with torch.cuda.stream(st):
            ops.bev_raster_batch(pts[g * n * N:(g + 1) * n * N], [i * N for i in range(n + 1)], par[:n], out=out[g * n:(g + 1) * n])
    torch.cuda.current_stream().wait_stream(s2)
for name, fn in (('one launch pair', one), ('2 groups / 2 streams', lambda: two(2)), ('4 groups / 2 streams', lambda: two(4))):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10): fn()
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 10 / TILES
    print(name, round(ms * 1e3, 2), 'us/tile', round((16 * N + 3 * 1152 * 1152 * 4) / ms / 1e6), 'GB/s')
