#!/usr/bin/env python3
"""Host PNG reader timings (csrc/png_reader.cpp over csrc/inflate.h): one 1152 x 1152 RGB tile on one thread - the synthetic bench tile as
PIL writes it, the same pixels through zlib level 6, a 90 %-empty tile - against zlib.decompress of the same stream, and a 16-file batch
on 1 / 4 / 8 / 16 threads.  CPU only."""
import io
import os
import struct
import sys
import tempfile
import time
import zlib

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lanemapping_amd import png_io, synth  # noqa: E402
from PIL import Image  # noqa: E402  (test data only)


def idat(data):
    pos, out = 8, []
    while pos < len(data):
        n, t = struct.unpack('>I', data[pos:pos + 4])[0], data[pos + 4:pos + 8]
        if t == b'IDAT':
            out.append(data[pos + 8:pos + 8 + n])
        pos += 12 + n
    return b''.join(out)


def best(f, reps=5):
    t = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        f()
        t = min(t, time.perf_counter() - t0)
    return t * 1e3


def chunk(t, d):
    return struct.pack('>I', len(d)) + t + d + struct.pack('>I', zlib.crc32(t + d) & 0xffffffff)


u8 = synth.bev_tile_u8(100, 1152)
rng = np.random.default_rng(1)
sparse = u8.copy()
sparse[rng.random((1152, 1152)) < 0.9] = 0
cases = {}
for name, img in (('synthetic tile, PIL', u8), ('90 % empty, PIL', sparse)):
    b = io.BytesIO()
    Image.fromarray(img).save(b, format='PNG')
    cases[name] = (b.getvalue(), img)
raw = zlib.decompress(idat(cases['synthetic tile, PIL'][0]))
cases['synthetic tile, zlib level 6'] = (b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', 1152, 1152, 8, 2, 0, 0, 0))
                                         + chunk(b'IDAT', zlib.compress(raw, 6)) + chunk(b'IEND', b''), u8)
for name, (png, img) in cases.items():
    z = idat(png)
    n = len(zlib.decompress(z))
    assert np.array_equal(png_io.decode_png(png), img)
    print(f'{name}: file {len(png) / 1e6:.2f} MB  decode_png {best(lambda: png_io.decode_png(png)):.2f} ms  '
          f'(own inflate {best(lambda: png_io.zlib_inflate(z, n)):.2f} ms, zlib.decompress {best(lambda: zlib.decompress(z)):.2f} ms, '
          f'PIL {best(lambda: np.array(Image.open(io.BytesIO(png)))):.2f} ms)')
d = tempfile.mkdtemp()
paths = []
for i in range(16):
    p = os.path.join(d, f'{i}.png')
    Image.fromarray(synth.bev_tile_u8(100 + i % 8, 1152)).save(p)
    paths.append(p)
out = np.empty((16, 1152, 1152, 3), np.uint8)
for th in (1, 4, 8, 16):
    t = best(lambda: png_io.read_png_batch(paths, threads=th, out=out), 4)
    print(f'{th} threads: 16 tiles in {t:.1f} ms = {16e3 / t:.0f} tiles/s')
