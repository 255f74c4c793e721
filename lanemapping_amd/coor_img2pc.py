"""BEV-image polylines -> point-cloud (LAS) frame, SURVEY §8f row f1.

Mirror of baseline/utils/coor_img2pc.py: `transform_coordinate_from_img_2_pc(params, img_seqs, img_seq_lens, bev_img)`
(:127-183) and the per-file driver `transform_coordinate_from_img_2_pc_single` (:185-220) with the same argument meaning,
outputs and file formats.  The arithmetic (empty-pixel elevation fill, affine, per-line least-squares elevation, quaternion
rotation q v q* / |q|, translation, las_read_offset) runs in liblanemap_hip.so (`lm_polyline_backproject`, host code in
csrc/backproject.cpp, double precision, bit-identical to the reference's numpy result); there is no Python fallback.
It is the inverse of the rasteriser's geometry (csrc/raster.hip), which closes the LAS -> BEV -> LAS round trip.
"""
import ctypes as C

import numpy as np

from ._lib import lib, check
from .io_utils import load_lane_seq, load_pc_2_img_transform_paras, save_seqs_json, save_seqs_txt


def transform_coordinate_from_img_2_pc(params, img_seqs, img_seq_lens, bev_img):
    """params: dict of `load_pc_2_img_transform_paras`; img_seqs [n_line, max_len, 2] (row, col); bev_img: HxWxC uint8 array
    (or anything np.array() turns into one, e.g. a PIL image).  Returns seqs_3d [n_line, max_len, 3] float64."""
    img_seqs = np.ascontiguousarray(np.asarray(img_seqs, dtype=np.float64)[:, :, :2])
    n_line, max_len, _ = img_seqs.shape
    img = np.array(bev_img)                                   # copy: the elevation fill edits the tile like the reference
    if img.ndim != 3 or img.dtype != np.uint8:
        raise ValueError(f'bev_img must be an HxWxC uint8 image, got {img.dtype} {img.shape}')
    img = np.ascontiguousarray(img)
    lens = np.ascontiguousarray(np.asarray(img_seq_lens, dtype=np.int32))
    if lens.shape[0] != n_line:
        raise ValueError('img_seq_lens must have one entry per line')
    p13 = np.array(list(params['img_reso'][:2]) + list(params['bev_img_offset'][:2]) +
                   [params['ele_reso'], params['local_min_ele']] + list(params['las_rotation_trans_quan'][:7]), dtype=np.float64)
    off = np.array(params['las_read_offset'][:3], dtype=np.float64)
    out = np.zeros((n_line, max_len, 3), dtype=np.float64)
    vp = C.c_void_p
    check(lib().lm_polyline_backproject(vp(img.ctypes.data), img.shape[0], img.shape[1], img.shape[2], vp(img_seqs.ctypes.data),
                                        vp(lens.ctypes.data), n_line, max_len, vp(p13.ctypes.data), vp(off.ctypes.data),
                                        vp(out.ctypes.data)))
    return out


def transform_coordinate_from_img_2_pc_single(img_seqfile_path, bev_img_path, pc_img_params_path, pc_seqfile_path,
                                              pc_seqfile_txt_path):
    """One tile: 2-D polyline JSON + BEV PNG + parameter file -> 3-D polyline JSON / TXT (reference :185-220)."""
    from .png_io import read_png
    img_seqs, img_seq_lens, _, _ = load_lane_seq(img_seqfile_path)
    if len(img_seqs) < 1:
        return
    params = load_pc_2_img_transform_paras(pc_img_params_path)
    pc_seqs = transform_coordinate_from_img_2_pc(params, img_seqs, img_seq_lens, read_png(bev_img_path))
    lines = []
    for i in range(pc_seqs.shape[0]):
        sub = pc_seqs[i, :img_seq_lens[i], :]
        lines.append({'seq': sub, 'seq_len': img_seq_lens[i], 'init_vertex': sub[0, :], 'end_vertex': sub[img_seq_lens[i] - 1, :]})
    save_seqs_json(lines, pc_seqfile_path)
    save_seqs_txt(lines, pc_seqfile_txt_path)
