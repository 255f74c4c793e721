"""LAS ingest without laspy: disk -> HBM, SURVEY §8f row f4.

`read_las(path, device)` mirrors baseline/datasets/laserlane_proposals.py:618-636 (`[N,4]` = x, y, z, intensity clipped to
[800, 33000] and normalised (i - 800) / 33000); `read_las_raw` keeps the raw intensity, which is what the rasteriser
(`ops.bev_raster_batch`) takes.  The public header block is parsed on the host (`lm_las_parse_header`); the point records
go to the GPU as raw bytes and are decoded there (`lm_las_decode_points`), straight into the `[N,4]` float32 layout of the
hot path.  Differences from the reference, both deliberate: float32 instead of float64 (the hot path is fp32; pass
`shift=las_read_offset` of the tile's parameter file so that metre-scale coordinates keep millimetre precision), and no
`exit()` on clouds with fewer than 5 points (a ValueError instead).
"""
import ctypes as C

import numpy as np
import torch

from ._lib import lib, check, LmLasHeader, LanemapHipError

INTEN_MIN, INTEN_MAX = 800.0, 33000.0


def parse_header(data):
    """bytes -> dict of the header fields (raises LanemapHipError on non-LAS / LAZ / truncated files)."""
    h = LmLasHeader()
    data = np.ascontiguousarray(np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data)
    check(lib().lm_las_parse_header(C.c_void_p(data.ctypes.data), int(data.shape[0]), C.byref(h)))
    return {'version': (h.version_major, h.version_minor), 'point_format': h.point_format, 'record_len': h.record_len,
            'n_points': h.n_points, 'offset_to_points': h.offset_to_points, 'scale': list(h.scale), 'offset': list(h.offset),
            'min': list(h.min_xyz), 'max': list(h.max_xyz)}


def decode_points(records_u8, record_len, n, scale, offset, shift=None, normalise=True, out=None):
    """records_u8: DEVICE uint8 tensor holding n records (padded to a multiple of 4 bytes) -> [n,4] float32 on that device
    (`out`: a contiguous [n,4] float32 tensor to decode into, e.g. a slice of a batch's point buffer)."""
    if not records_u8.is_cuda:
        raise LanemapHipError('las_io.decode_points needs the records on an MI355X (HIP) device; no CPU fallback exists')
    if out is None:
        out = torch.empty((n, 4), device=records_u8.device, dtype=torch.float32)
    elif tuple(out.shape) != (n, 4) or out.dtype != torch.float32 or not out.is_contiguous() or out.device != records_u8.device:
        raise ValueError(f'las_io.decode_points: out must be a contiguous [{n},4] float32 tensor on {records_u8.device}')
    d3 = lambda v: (C.c_double * 3)(*[float(x) for x in v])
    check(lib().lm_las_decode_points(C.c_void_p(torch._C._cuda_getCurrentRawStream(records_u8.device.index)), C.c_void_p(records_u8.data_ptr()),
                                     int(record_len), int(n), d3(scale), d3(offset), d3(shift) if shift is not None else None,
                                     INTEN_MIN, INTEN_MAX, int(normalise), C.c_void_p(out.data_ptr())))
    return out


def _read(path, device, shift, normalise):
    data = np.fromfile(path, dtype=np.uint8)
    h = parse_header(data)
    n, rl, off = h['n_points'], h['record_len'], h['offset_to_points']
    nbytes = n * rl
    padded = np.zeros(((nbytes + 3) // 4 * 4,), dtype=np.uint8)
    padded[:nbytes] = data[off:off + nbytes]
    rec = torch.from_numpy(padded).to(device, non_blocking=True)
    return decode_points(rec, rl, n, h['scale'], h['offset'], shift, normalise), h


def read_las(filepath, device='cuda:0', shift=None):
    """-> [N,4] float32 (x, y, z, normalised intensity) on `device`, like the reference's read_las."""
    pts, _ = _read(filepath, torch.device(device), shift, True)
    if pts.shape[0] < 5:
        raise ValueError(f'{filepath}: only {pts.shape[0]} lidar points')
    return pts


def read_las_raw(filepath, device='cuda:0', shift=None):
    """-> ([N,4] float32 with RAW intensity, header dict): the record layout lm_bev_raster_batch consumes."""
    return _read(filepath, torch.device(device), shift, False)
