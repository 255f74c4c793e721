"""Host (C++) tail of the hot path, called through the C-ABI with NumPy buffers.

  cluster_endpoints  -> lm_endp_cluster      (reference heads/polyline_fpn_vit_vertex_2.py:661-688, :903-924)
  assemble_polylines -> lm_polyline_assemble (reference :805-861 + utils/polyline_utils.py)
"""
import ctypes as C

import numpy as np

from ._lib import lib, check

IMG = 1152


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def cluster_endpoints(topk_idx, crop_w=IMG - 40, clip=20, k0=240, k_step=10, k_max=500, radius=20, min_clusters=4):
    """topk_idx: int32 flat indices into the cropped score map, best first. -> ([n,2] (h,w) int32, K used)."""
    idx = np.ascontiguousarray(topk_idx, dtype=np.int32)
    out = np.zeros((len(idx), 2), dtype=np.int32)
    n_out = C.c_int(0)
    k_used = C.c_int(0)
    check(lib().lm_endp_cluster(_p(idx), len(idx), crop_w, clip, k0, k_step, k_max, radius, min_clusters,
                                _p(out), len(out), C.addressof(n_out), C.addressof(k_used)))
    return out[:n_out.value].copy(), k_used.value


def assemble_polylines(prop_conf, prop_v_ext, cls_offset, bi_seg_rows, endp_hw, obj_thre=0.3, min_vertices=8):
    """One tile: prop_conf [P,2] f32, prop_v_ext [P,R] f32, cls_offset [P,R] f64, bi_seg_rows [R,1152] f32
    (image rows 3::8), endp_hw [n,2] int32 -> (lanes [P,R,2] f64 (col px | -1, semantic), kept endpoints [m,2])."""
    pc = np.ascontiguousarray(prop_conf, dtype=np.float32)
    ve = np.ascontiguousarray(prop_v_ext, dtype=np.float32)
    co = np.ascontiguousarray(cls_offset, dtype=np.float64)
    rows = np.ascontiguousarray(bi_seg_rows, dtype=np.float32)
    ep = np.ascontiguousarray(endp_hw, dtype=np.int32).reshape(-1, 2)
    P, R = ve.shape
    assert pc.shape == (P, 2) and co.shape == (P, R) and rows.shape == (R, IMG)
    lanes = np.empty((P, R, 2), dtype=np.float64)
    keep = np.ones(max(len(ep), 1), dtype=np.int32)
    check(lib().lm_polyline_assemble(_p(pc), _p(ve), _p(co), _p(rows), _p(ep), len(ep), P, R, obj_thre, min_vertices,
                                     _p(lanes), _p(keep)))
    return lanes, ep[keep[:len(ep)] > 0].copy()


def raster_semantic_map(lanes):
    """[P,R,2] lanes -> [1152,1152] f64 map with 1 (solid) / 2 (dashed) polylines (reference renew_semantic_map)."""
    lanes = np.ascontiguousarray(lanes, dtype=np.float64)
    out = np.empty((IMG, IMG), dtype=np.uint8)
    check(lib().lm_raster_polylines(_p(lanes), lanes.shape[0], lanes.shape[1], _p(out)))
    return out.astype(np.float64)


def trace_lines(cols, seg_rows=None):
    """[n,R] f64 column px (<= 0 = none) -> traced / merged / gap-filled lines [n,R] (reference smooth_cls_line_per_batch)."""
    c = np.ascontiguousarray(cols, dtype=np.float64)
    out = np.empty_like(c)
    seg = None if seg_rows is None else np.ascontiguousarray(seg_rows, dtype=np.float32)
    check(lib().lm_trace_lines(_p(c), c.shape[0], c.shape[1], None if seg is None else _p(seg), _p(out)))
    return out
