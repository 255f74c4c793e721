"""Per-tile polyline output / tile parameter input, format-compatible with the reference.

  save_lane_seq_2d               <- baseline/utils/io_utils.py:58-93 (+ save_seqs_json :11-15, save_seqs_txt :17-26)
  save_seqs_json/txt/list        <- baseline/utils/io_utils.py:11-56 (3-D polylines after the back-projection / merge)
  load_lane_seq                  <- baseline/utils/io_utils.py:100-123
  load_pc_2_img_transform_paras  <- baseline/utils/io_utils.py:125-150 (values on lines 1,3,5,...,13 of the file)
  pack_lane_vertices             <- heads/polyline_fpn_vit_vertex_2.py:997-1000 (row = 3 + 8 i, col, semantic)
The JSON text equals the reference's byte for byte (json.dump(indent=4) of python floats / ints).
"""
import json
import os

import numpy as np


def pack_lane_vertices(lanes, row_size=144):
    v = np.zeros((lanes.shape[0], row_size, 3))
    v[:, :, 0] = np.arange(3, row_size * 8, 8)
    v[:, :, 1:] = lanes
    return v


def lane_records(lane_vertexes, with_pervertex_semantics=True):
    recs = []
    for lane in np.asarray(lane_vertexes):
        pv = lane[lane[:, 1] > 0]
        if pv.shape[0] < 2:
            continue
        if not with_pervertex_semantics:
            pv = pv[:, :-1]
        recs.append({'seq_len': int(pv.shape[0]), 'seq': pv.tolist(), 'init_vertex': pv[0].tolist(),
                     'end_vertex': pv[-1].tolist()})
    return recs


def lane_json_text(lane_vertexes, with_pervertex_semantics=True):
    """Text of json.dumps(lane_records(lane_vertexes, ...), indent=4), byte for byte, from the library's writer
    (csrc/lane_json.cpp): json's indented mode runs its pure-Python encoder, 13-30 ms per tile."""
    import ctypes as C
    from ._lib import lib, LanemapHipError
    v = np.ascontiguousarray(lane_vertexes, dtype=np.float64)
    assert v.ndim == 3 and v.shape[2] == 3
    args = (C.c_void_p(v.ctypes.data), v.shape[0], v.shape[1], int(bool(with_pervertex_semantics)))
    n = lib().lm_lane_json_text(*args, None, 0)
    if n < 0:
        raise LanemapHipError(lib().lm_last_error().decode())
    buf = C.create_string_buffer(n + 1)
    lib().lm_lane_json_text(*args, buf, n + 1)
    return buf.value.decode()


def save_lane_seq_2d(lane_vertexes, lane_seq_path, with_pervertex_semantics=True):
    if os.path.splitext(lane_seq_path)[1] == '.txt':
        recs = lane_records(lane_vertexes, with_pervertex_semantics)
        with open(lane_seq_path, 'w') as f:
            for i, line in enumerate(recs):
                for vtx in line['seq']:
                    f.write(' '.join(str(item) for item in vtx) + ' ' + str(i) + '\n')
    else:                                                       # == json.dump(recs, f, indent=4), byte for byte
        import ctypes as C
        from ._lib import lib, check
        v = np.ascontiguousarray(lane_vertexes, dtype=np.float64)
        assert v.ndim == 3 and v.shape[2] == 3
        check(lib().lm_lane_json_write(C.c_void_p(v.ctypes.data), v.shape[0], v.shape[1], int(bool(with_pervertex_semantics)),
                                       str(lane_seq_path).encode()))


class NpEncoder(json.JSONEncoder):
    """data/convert_data.py:15-23: numpy scalars / arrays -> JSON numbers / lists."""

    def default(self, obj):
        if isinstance(obj, np.integer):
            return int(obj)
        if isinstance(obj, np.floating):
            return float(obj)
        if isinstance(obj, np.ndarray):
            return obj.tolist()
        return super().default(obj)


_SEQ_KEYS = ['seq', 'seq_len', 'init_vertex', 'end_vertex']


def save_seqs_json(seq_list, seq_path):
    """baseline/utils/io_utils.py:11-15: json.dump(seq_list, indent=4, cls=NpEncoder).  Records of the polyline schema (the keys above,
    in that order, `seq` a float [n, D] array whose first / last rows are the two vertices) are written by the library's JSON writer,
    byte-identical and ~20x faster than CPython's indented encoder; anything else goes through json.dump."""
    native = len(seq_list) > 0
    for r in seq_list:
        if not (isinstance(r, dict) and list(r) == _SEQ_KEYS):
            native = False
            break
        sq = np.asarray(r['seq'])
        if not (sq.ndim == 2 and sq.dtype == np.float64 and sq.shape[0] >= 1 and sq.shape[0] == int(r['seq_len'])
                and sq.shape[1] == np.shape(r['init_vertex'])[0] == np.shape(r['end_vertex'])[0]
                and np.array_equal(sq[0], np.asarray(r['init_vertex'], dtype=np.float64), equal_nan=True)
                and np.array_equal(sq[-1], np.asarray(r['end_vertex'], dtype=np.float64), equal_nan=True)
                and sq.shape[1] == np.asarray(seq_list[0]['seq']).shape[1]
                and isinstance(r['seq_len'], (int, np.integer)) and not isinstance(r['seq_len'], bool)):
            native = False
            break
    if not native:
        with open(seq_path, 'w') as f:
            json.dump(seq_list, f, indent=4, cls=NpEncoder)
        return
    import ctypes as C
    from ._lib import lib, check
    lens = np.array([int(r['seq_len']) for r in seq_list], dtype=np.int32)
    D = np.asarray(seq_list[0]['seq']).shape[1]
    seqs = np.zeros((len(seq_list), int(lens.max()), D))
    for i, r in enumerate(seq_list):
        seqs[i, :lens[i]] = r['seq']
    check(lib().lm_seqs_json_write(C.c_void_p(seqs.ctypes.data), C.c_void_p(lens.ctypes.data), len(seq_list), seqs.shape[1], D,
                                   str(seq_path).encode()))


def save_seqs_txt(seq_list, seq_path):
    """baseline/utils/io_utils.py:17-26: one vertex per line, trailing line id."""
    with open(seq_path, 'w') as f:
        for i, seq in enumerate(seq_list):
            for vtx in seq['seq']:
                f.write(' '.join(str(item) for item in vtx) + ' ' + str(i) + '\n')


def save_seqs_list(lane_vertexes, lane_seq_path):
    """baseline/utils/io_utils.py:28-56: list of [n_i, d] arrays -> records of the lines with >= 2 vertices."""
    lines = [{'seq_len': v.shape[0], 'seq': v, 'init_vertex': v[0, :], 'end_vertex': v[-1, :]}
             for v in lane_vertexes if v.shape[0] >= 2]
    if os.path.splitext(lane_seq_path)[1] == '.txt':
        save_seqs_txt(lines, lane_seq_path)
    else:
        save_seqs_json(lines, lane_seq_path)


def load_lane_seq(seqfile_path, dim_coor=2):
    with open(seqfile_path) as f:
        data = json.load(f)
    seq_lens = [a['seq_len'] for a in data]
    init_points = [a['init_vertex'] for a in data]
    end_points = [a['end_vertex'] for a in data]
    if len(seq_lens) < 2:          # reference quirk: a single line yields an empty list
        return [], seq_lens, init_points, end_points
    seq = np.zeros((len(seq_lens), max(seq_lens), dim_coor))
    for i, a in enumerate(data):
        if seq_lens[i]:
            seq[i, :seq_lens[i]] = [x[0:dim_coor] for x in a['seq']]
    return seq, seq_lens, init_points, end_points


def load_pc_2_img_transform_paras(param_path):
    with open(param_path) as f:
        lines = f.read().split('\n')
    fl = lambda s: [float(t) for t in s.split(' ')]
    return {'coor_las_path': lines[1], 'las_read_offset': fl(lines[3]), 'las_rotation_trans_quan': fl(lines[5]),
            'bev_img_offset': fl(lines[7]), 'img_reso': fl(lines[9]), 'local_min_ele': float(lines[11]),
            'ele_reso': float(lines[13])}


def raster_params_from_file(param_path):
    """Reference parameter file -> LmRasterParams for lm_bev_raster_batch (points must already have
    `las_read_offset` subtracted, as the reference's LAS reader hands them over)."""
    from .ops import make_raster_params
    p = load_pc_2_img_transform_paras(param_path)
    q = p['las_rotation_trans_quan']
    return make_raster_params(quat=q[3:7], trans=q[0:3], bev_img_offset=p['bev_img_offset'], img_reso=p['img_reso'],
                              local_min_ele=p['local_min_ele'], ele_reso=p['ele_reso'])


# ------------------------------------------------------------------------------------------------ label JSON (training-set annotations)
def load_label_seq(seq_path):
    """Annotation file of one tile (data/convert_data.py:25-52 `load_seq`): a list of areas {'seq': [[x, y, ...], ...], 'init_vertex',
    'end_vertex', 'semantic', 'instance'} -> (seq [n, max_len, 2] zero padded, seq_lens, semantics, instances, init points, end points)."""
    with open(seq_path) as f:
        data = json.load(f)
    seq_lens = [len(a['seq']) for a in data]
    init_points = [a['init_vertex'] for a in data]
    end_points = [a['end_vertex'] for a in data]
    semantics = [a['semantic'] for a in data]
    instances = [a['instance'] for a in data]
    seq = np.zeros((len(seq_lens), max(seq_lens), 2))          # (an empty file raises ValueError in the reference as well: max of [])
    for i, a in enumerate(data):
        seq[i, :seq_lens[i]] = [v[0:2] for v in a['seq']]
    return seq, seq_lens, semantics, instances, init_points, end_points


def save_label_seq(seqs, seq_lens, seqs_semantic, seqs_instance, seqs_orient, seqs_filename):
    """Inverse (data/convert_data.py:54-70 `save_seq`): one dictionary per line with the keys and key order of the reference,
    json.dump without indentation through NpEncoder."""
    lines = []
    for i, n in enumerate(seq_lens):
        lines.append({'semantic': seqs_semantic[i], 'instance': seqs_instance[i], 'seq_len': n, 'seq': seqs[i, :n, :],
                      'init_vertex': seqs[i, 0, :], 'end_vertex': seqs[i, n - 1, :], 'seq_orient': seqs_orient[i, :n]})
    with open(seqs_filename, 'w') as f:
        json.dump(lines, f, cls=NpEncoder)
