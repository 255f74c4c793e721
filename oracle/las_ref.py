"""Oracle (TEST INFRASTRUCTURE ONLY) for SURVEY §8f row f4: LAS files.

The reference reads LAS through laspy (baseline/datasets/laserlane_proposals.py:618-636), which is not installed here and
not vendored => PARITY vs laspy UNPINNED; this module restates the ASPRS LAS 1.2 / 1.4 layout (public header block, point
data record formats 0-3 and 6) with numpy, both as a WRITER (to synthesise test files) and as a READER
(`read_las_ref` = the reference's read_las arithmetic: X * scale + offset in float64, intensity clip/normalise).
"""
import struct

import numpy as np

RECORD_LEN = {0: 20, 1: 28, 2: 26, 3: 34, 6: 30}


def write_las(path, xyz, intensity, point_format=1, version=(1, 2), scale=(0.001, 0.001, 0.001), offset=(0., 0., 0.), extra_bytes=0,
              vlr_bytes=0):
    """Write an uncompressed LAS file; coordinates are quantised to int32 with the given scale/offset."""
    xyz = np.asarray(xyz, np.float64)
    n = xyz.shape[0]
    q = np.round((xyz - np.asarray(offset)) / np.asarray(scale)).astype(np.int32)
    rl = RECORD_LEN[point_format] + extra_bytes
    rec = np.zeros((n, rl), np.uint8)
    rec[:, 0:12] = q.view(np.uint8).reshape(n, 12)
    rec[:, 12:14] = np.asarray(intensity, np.uint16).reshape(n, 1).view(np.uint8).reshape(n, 2)
    rec[:, 14:rl] = (np.arange(n)[:, None] * 7 + np.arange(rl - 14)[None, :]) % 251        # junk in the fields we must skip
    hs = 375 if version[1] >= 4 else 227
    h = bytearray(hs)
    h[0:4] = b'LASF'
    h[24], h[25] = version
    h[26:26 + 9] = b'lanemap-t'
    struct.pack_into('<H', h, 94, hs)
    struct.pack_into('<I', h, 96, hs + vlr_bytes)
    struct.pack_into('<I', h, 100, 0)
    h[104] = point_format
    struct.pack_into('<H', h, 105, rl)
    struct.pack_into('<I', h, 107, n if version[1] < 4 else 0)        # 1.4 writers may leave the legacy count at 0
    struct.pack_into('<3d', h, 131, *scale)
    struct.pack_into('<3d', h, 155, *offset)
    real = q * np.asarray(scale) + np.asarray(offset)
    for a in range(3):
        struct.pack_into('<2d', h, 179 + 16 * a, real[:, a].max() if n else 0., real[:, a].min() if n else 0.)
    if version[1] >= 4:
        struct.pack_into('<Q', h, 247, n)
    with open(path, 'wb') as f:
        f.write(bytes(h))
        f.write(bytes(vlr_bytes))
        f.write(rec.tobytes())
    return q


def read_las_ref(path, shift=None, normalise=True):
    """numpy reader: -> [N,4] float64 (x, y, z, intensity) with read_las's arithmetic."""
    data = np.fromfile(path, np.uint8)
    assert bytes(data[0:4]) == b'LASF'
    minor = int(data[25])
    off = struct.unpack_from('<I', data, 96)[0]
    rl = struct.unpack_from('<H', data, 105)[0]
    n = struct.unpack_from('<I', data, 107)[0]
    if minor >= 4:
        n64 = struct.unpack_from('<Q', data, 247)[0]
        n = n64 if n64 > 0 else n
    scale = np.array(struct.unpack_from('<3d', data, 131))
    offset = np.array(struct.unpack_from('<3d', data, 155))
    rec = data[off:off + n * rl].reshape(n, rl)
    q = rec[:, 0:12].copy().view(np.int32).reshape(n, 3)
    inten = rec[:, 12:14].copy().view(np.uint16).reshape(n).astype(np.float64)
    sh = np.zeros(3) if shift is None else np.asarray(shift, np.float64)
    xyz = q * scale + (offset - sh)
    if normalise:
        inten = (np.clip(inten, 800.0, 33000.0) - 800.0) / 33000.0
    return np.concatenate([xyz, inten[:, None]], axis=1)
