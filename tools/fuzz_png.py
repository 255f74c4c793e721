#!/usr/bin/env python3
"""Offline fuzz of the PNG tile reader against PIL: small random images of every supported mode, 20 mutations each (bit flips, chunk CRCs
repaired in 70 % of the cases so that the zlib / unfilter paths are reached).  Wherever both readers accept a file the pixels must be equal; PIL
accepts more (it tolerates a wrong Adler-32, 16-bit depth, trailing data ...).  usage: fuzz_png.py   (90 s, CPU only)"""
import os
import sys, zlib, io, struct, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from lanemapping_amd import png_io
from lanemapping_amd._lib import LanemapHipError
warnings.simplefilter('ignore')
rng = np.random.default_rng(3)
def fix_crcs(data):
    out, pos = bytearray(data[:8]), 8
    while pos + 12 <= len(data):
        n = struct.unpack('>I', data[pos:pos + 4])[0]
        if pos + 12 + n > len(data): break
        body = data[pos + 4:pos + 8 + n]
        out += data[pos:pos + 4] + body + struct.pack('>I', zlib.crc32(body) & 0xffffffff)
        pos += 12 + n
    return bytes(out + data[pos:])
t0 = time.time(); both = mine_only = pil_only = neither = diff = 0
while time.time() - t0 < 90:
    mode = rng.choice(['RGB', 'RGBA', 'L', 'LA'])
    h, w = int(rng.integers(1, 60)), int(rng.integers(1, 80))
    ch = {'RGB': 3, 'RGBA': 4, 'L': 1, 'LA': 2}[mode]
    img = rng.integers(0, 256, (h, w, ch), dtype=np.uint8)
    if rng.random() < 0.5: img[..., 0] = (np.add.outer(np.arange(h), np.arange(w)) % 251).astype(np.uint8)
    b = io.BytesIO(); Image.fromarray(img[..., 0] if ch == 1 else img, mode).save(b, 'PNG', compress_level=int(rng.integers(0, 10))); good = b.getvalue()
    for _ in range(20):
        d = bytearray(good)
        for _ in range(int(rng.integers(1, 4))): d[int(rng.integers(8, len(d)))] ^= 1 << int(rng.integers(0, 8))
        d = fix_crcs(bytes(d)) if rng.random() < 0.7 else bytes(d)
        try: a = png_io.decode_png(d)
        except LanemapHipError: a = None
        try:
            im = Image.open(io.BytesIO(d)); im.load(); p = np.array(im) if im.mode in ('RGB', 'RGBA', 'L', 'LA') else 'mode'
        except Exception: p = None
        if a is not None and p is not None and not isinstance(p, str):
            both += 1
            if a.shape != p.shape or not np.array_equal(a, p): diff += 1; print('DIFF', a.shape, p.shape)
        elif a is not None: mine_only += 1
        elif p is not None: pil_only += 1
        else: neither += 1
print(f'both accept {both} (pixel differences: {diff}), only this reader {mine_only}, only PIL {pil_only}, neither {neither}')
