#!/usr/bin/env python3
"""Offline fuzz of the library's DEFLATE decoder (csrc/inflate.h through lm_zlib_inflate) against zlib.decompress: random streams of
every level / strategy / window / memory level, 30 mutations each (truncation, bit flips, byte overwrites, damage with a repaired Adler-32 so that
the body is really decoded).  Both must refuse, or both accept with identical bytes (zlib.decompress ignores bytes after the end of the stream,
this decoder refuses them: printed as a divergence, the only kind seen).  usage: fuzz_inflate.py [seed] [seconds]   (CPU only)"""
import os
import sys, zlib, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from lanemapping_amd import png_io
from lanemapping_amd._lib import LanemapHipError
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60
def mk():
    kind = rng.integers(0, 6)
    n = int(rng.integers(0, 5000))
    if kind == 0: raw = rng.integers(0, 256, n, dtype=np.uint8).tobytes()
    elif kind == 1: raw = bytes(n)
    elif kind == 2: raw = (rng.integers(0, 4, n, dtype=np.uint8) * 60).tobytes()
    elif kind == 3: raw = bytes(rng.integers(0, 256, int(rng.integers(1, 40)), dtype=np.uint8)) * (n // 20)
    elif kind == 4: raw = np.minimum(rng.geometric(0.2, n), 255).astype(np.uint8).tobytes()
    else: raw = b'abc' * (n // 3) + rng.integers(0, 256, 50, dtype=np.uint8).tobytes()
    c = zlib.compressobj(int(rng.integers(0, 10)), zlib.DEFLATED, int(rng.integers(9, 16)), int(rng.integers(1, 10)), int(rng.choice([0, 1, 2, 3, 4])))
    return raw, c.compress(raw) + c.flush()
t0 = time.time(); n = same = refused = diverge = 0
while time.time() - t0 < budget:
    raw, z = mk()
    assert png_io.zlib_inflate(z, len(raw) + 10) == raw
    for _ in range(30):
        d = bytearray(z)
        m = rng.integers(0, 4)
        if m == 0 and len(d) > 2: d = d[:int(rng.integers(0, len(d)))]
        elif m == 1:
            for _ in range(int(rng.integers(1, 4))): d[int(rng.integers(0, len(d)))] ^= 1 << int(rng.integers(0, 8))
        elif m == 2:
            for _ in range(int(rng.integers(1, 3))): d[int(rng.integers(0, len(d)))] = int(rng.integers(0, 256))
        else:
            # corrupt but repair the adler so that the body is really decoded
            for _ in range(int(rng.integers(1, 3))): d[int(rng.integers(2, max(3, len(d) - 4)))] ^= 1 << int(rng.integers(0, 8))
            try:
                dd = zlib.decompressobj(-15); body = dd.decompress(bytes(d[2:-4]), 1 << 20)
                if dd.eof and not dd.unused_data: d[-4:] = zlib.adler32(body).to_bytes(4, 'big')
            except zlib.error: pass
        d = bytes(d)
        try: want = zlib.decompress(d)
        except zlib.error: want = None
        try: got = png_io.zlib_inflate(d, 1 << 20)
        except LanemapHipError: got = None
        n += 1
        if want is None and got is None: refused += 1
        elif want is not None and got == want: same += 1
        else:
            diverge += 1
            if diverge <= 5: print('DIVERGE', want is None, got is None, len(d), d[:40].hex())
print(f'{n} mutated streams: {same} accepted with identical bytes, {refused} refused by both, {diverge} divergences')
