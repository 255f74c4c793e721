"""PNG tile ingest without PIL, SURVEY §8f row f4.

`read_png(path)` returns what the reference's `load_img` starts from, `np.array(Image.open(path))`
(baseline/datasets/laserlane_proposals.py:85-98): uint8 [H,W] for greyscale files, [H,W,C] otherwise.  `read_png_batch` inflates a
homogeneous batch of tiles on host threads (zlib releases no GIL in PIL; here the whole decode runs outside Python) into one
[n,H,W,C] array, the layout `ops.tile_ingest` (`lm_tile_ingest_u8`) takes.  The decoder lives in the C-ABI library
(`csrc/png_reader.cpp` over `csrc/inflate.h`, no zlib): 8-bit, non-interlaced grey / grey+alpha / RGB / RGBA; anything else raises
LanemapHipError.
"""
import ctypes as C

import numpy as np

from ._lib import lib, check


def png_info(data):
    """bytes / uint8 array of a PNG file -> (H, W, channels)."""
    buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    h, w, c = C.c_int(), C.c_int(), C.c_int()
    check(lib().lm_png_info(C.c_void_p(buf.ctypes.data), int(buf.shape[0]), C.byref(h), C.byref(w), C.byref(c)))
    return h.value, w.value, c.value


def decode_png(data):
    """bytes / uint8 array of a PNG file -> uint8 [H,W] (greyscale) or [H,W,C]."""
    buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    h, w, c = png_info(buf)
    out = np.empty((h, w, c), dtype=np.uint8)
    check(lib().lm_png_decode_u8(C.c_void_p(buf.ctypes.data), int(buf.shape[0]), C.c_void_p(out.ctypes.data), int(out.nbytes)))
    return out[:, :, 0] if c == 1 else out


def zlib_inflate(data, capacity):
    """zlib stream (RFC 1950) -> bytes, through the library's own DEFLATE decoder (csrc/inflate.h, the one the PNG reader runs on).
    `capacity` bounds the output: a stream that inflates to more raises LanemapHipError."""
    buf = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data, dtype=np.uint8)
    out = np.empty(max(int(capacity), 1), dtype=np.uint8)
    got = C.c_int64()
    check(lib().lm_zlib_inflate(C.c_void_p(buf.ctypes.data if buf.size else out.ctypes.data), int(buf.shape[0]), C.c_void_p(out.ctypes.data),
                                int(capacity), C.byref(got)))
    return out[:got.value].tobytes()


def read_png(path):
    with open(path, 'rb') as f:
        return decode_png(f.read())


def read_png_batch(paths, threads=8, out=None):
    """List of PNG files of identical size / channel count -> uint8 [n,H,W,C] (C kept even when 1).  `out` may be a preallocated
    (e.g. pinned) uint8 array / tensor-backed array of that shape."""
    paths = [str(p) for p in paths]
    if not paths:
        raise ValueError('read_png_batch: no files')
    with open(paths[0], 'rb') as f:
        h, w, c = png_info(f.read(64))
    if out is None:
        out = np.empty((len(paths), h, w, c), dtype=np.uint8)
    assert out.dtype == np.uint8 and out.shape == (len(paths), h, w, c) and out.flags['C_CONTIGUOUS']
    arr = (C.c_char_p * len(paths))(*[p.encode() for p in paths])
    check(lib().lm_png_decode_files_u8(arr, len(paths), C.c_void_p(out.ctypes.data), h, w, c, max(1, int(threads))))
    return out
