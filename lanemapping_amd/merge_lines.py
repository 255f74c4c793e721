"""Cross-tile merge of LAS-frame polylines into map-level lines: ctypes marshalling over the host C++
merger of the C-ABI (csrc/merge_lines.cpp).  Same entry points as baseline/utils/merge_lines.py:

  merge_lines(seq_filenames)   <- :166-291  (sorted per-tile JSON files -> list of merged [n, 3] arrays)
  downsample_seqs(seq, 0.6)    <- :133-153
  LineMerger                   streaming form: add_tile(list of [n,3] arrays) per tile, finish() -> merged lines
"""
import ctypes as C

import numpy as np

from ._lib import lib, check, LanemapHipError
from .io_utils import load_lane_seq


class LineMerger:
    def __init__(self):
        self._h = C.c_void_p(lib().lm_merge_create())
        if not self._h:
            raise LanemapHipError('lm_merge_create failed')

    def add_tile(self, lines):
        """lines: list of [n_i >= 2, 3] arrays (x, y, z in the LAS frame) of ONE tile, tiles in sorted file-name order."""
        lens = np.ascontiguousarray([len(l) for l in lines], dtype=np.int32)
        pts = (np.ascontiguousarray(np.concatenate([np.asarray(l, dtype=np.float64).reshape(-1, 3) for l in lines]))
               if len(lines) else np.zeros((0, 3)))
        code = lib().lm_merge_add_tile(self._h, pts.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p), len(lines))
        if code != 0:
            msg = lib().lm_last_error().decode()
            if 'IndexError' in msg:
                raise IndexError(msg)              # the reference fails the same way on this input
            check(code)

    def finish(self):
        tot = C.c_long(0)
        n = lib().lm_merge_finish(self._h, C.byref(tot))
        lens = np.zeros(max(n, 1), dtype=np.int32)
        pts = np.zeros((max(tot.value, 1), 3))
        check(lib().lm_merge_result(self._h, pts.ctypes.data_as(C.c_void_p), lens.ctypes.data_as(C.c_void_p)))
        out, at = [], 0
        for i in range(n):
            out.append(pts[at:at + lens[i]].copy())
            at += lens[i]
        return out

    def __del__(self):
        h, self._h = getattr(self, '_h', None), None
        if h:
            lib().lm_merge_destroy(h)


def merge_lines(seq_filenames, verbose=False):
    """List of per-tile 3-D polyline JSON files -> list of merged [n, 3] arrays."""
    m = LineMerger()
    for name in sorted(seq_filenames):
        seqs, lens, _, _ = load_lane_seq(name, dim_coor=3)
        m.add_tile([s[:lens[i]] for i, s in enumerate(seqs)])
    return m.finish()


def downsample_seqs(seq_base, dist_min=0.6):
    seq = np.ascontiguousarray(seq_base, dtype=np.float64).reshape(-1, 3)
    out = np.zeros((len(seq) + 1, 3))
    n = lib().lm_downsample_seq(seq.ctypes.data_as(C.c_void_p), len(seq), float(dist_min), out.ctypes.data_as(C.c_void_p))
    return out[:n].copy()
