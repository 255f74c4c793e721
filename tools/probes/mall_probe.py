#!/usr/bin/env python3
"""Does the 256 MiB Infinity Cache absorb a write -> read round trip?  (decides whether the rasteriser's 16.8 MB/tile
of band records can stay on die between its two passes)"""
import torch

dev = torch.device('cuda:0')
big = torch.empty(1 << 28, device=dev)          # 1 GiB flusher


def t(fn, rep=20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for _ in range(rep):
        a.record()
        fn()
        b.record()
        torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best


for mb in (16, 32, 64, 128, 192, 256, 512, 1024):
    n = mb * (1 << 20) // 4
    x = torch.empty(n, device=dev)
    y = torch.empty(n, device=dev)
    x.fill_(1.0)
    rd = t(lambda: x.sum())                       # re-read of the same buffer
    def flush_read():
        big.fill_(0.0)
    wr = t(lambda: x.fill_(2.0))
    cp = t(lambda: y.copy_(x))
    # write then read back-to-back
    def wr_rd():
        x.fill_(3.0)
        x.sum()
    both = t(wr_rd)
    print(f'{mb:5d} MB: read {mb / 1e3 / rd * 1e3 / 1e3:6.2f} TB/s  write {mb / 1e3 / wr:6.2f} TB/s  copy(r+w) {2 * mb / 1e3 / cp:6.2f} TB/s  '
          f'write+read {2 * mb / 1e3 / both:6.2f} TB/s  ({rd * 1e3:.1f}/{wr * 1e3:.1f}/{cp * 1e3:.1f}/{both * 1e3:.1f} us)')
