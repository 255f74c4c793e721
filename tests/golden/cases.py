"""Seeded test-case builders shared by make_golden.py (reference side) and tests/ (build side).

Everything is derived from integer seeds through lanemapping_amd.synth, so the golden .npz
files only need to carry expected outputs.
"""
import os
import sys

import json

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from lanemapping_amd import synth  # noqa: E402

IMG = 1152
NUM_POSTPROC_CASES = 24


def vit_input(seed):
    return synth.normalish(seed, 64 * 144 * 144).astype(np.float32).reshape(1, 64, 144, 144)


def head_inputs(seed, batch=1):
    x = synth.normalish(seed, batch * 8 * 144 * 144, 1).astype(np.float32).reshape(batch, 8, 144, 144)
    x_up = synth.normalish(seed, batch * 8 * 288 * 288, 2).astype(np.float32).reshape(batch, 8, 288, 288)
    return x, x_up


def _blobs(seed, n_blob, stream):
    par = synth.uniform(seed, n_blob * 3, stream)
    ys = 60 + par[0::3] * (IMG - 120)
    xs = 60 + par[1::3] * (IMG - 120)
    amp = 6.0 + 2.0 * par[2::3]
    return ys, xs, amp


def decode_inputs(seed, batch=1):
    """Synthetic raw network outputs with planted structure (decode golden G5/G7)."""
    out = {}
    out['proposal_conf'] = (2.0 * synth.normalish(seed, batch * 72 * 2, 1)).astype(np.float32).reshape(batch, 72, 2)
    out['ext2'] = (2.0 * synth.normalish(seed, batch * 72 * 144 * 3, 2)).astype(np.float32).reshape(batch, 72, 144, 3)
    out['cls2'] = (2.0 * synth.normalish(seed, batch * 72 * 144 * 10, 3)).astype(np.float32).reshape(batch, 72, 144, 10)
    out['offset2'] = (0.3 * synth.normalish(seed, batch * 72 * 144 * 10, 4)).astype(np.float32).reshape(batch, 72, 144, 10)
    out['orient'] = (2.0 * synth.normalish(seed, batch * 11 * 144 * 144, 5)).astype(np.float32).reshape(batch, 11, 144, 144)
    rows = np.arange(IMG, dtype=np.float64)[:, None]
    cols = np.arange(IMG, dtype=np.float64)[None, :]
    sem = np.zeros((batch, 3, IMG, IMG), dtype=np.float32)
    endp = np.zeros((batch, 1, IMG, IMG), dtype=np.float32)
    for b in range(batch):
        noise = synth.normalish(seed, 3 * IMG * IMG, 10 + b).reshape(3, IMG, IMG)
        s = 0.4 * noise
        s[0] += 2.0
        par = synth.uniform(seed, 32, 20 + b)
        for k in range(6):
            centre = (0.1 + 0.15 * k + 0.03 * par[k]) * IMG + 0.1 * (par[8 + k] - 0.5) * rows
            near = np.exp(-((cols - centre) ** 2) / (2 * 2.5 ** 2))
            s[1 + (k % 2)] += 4.0 * near
        sem[b] = s.astype(np.float32)
        e = -4.0 + 0.25 * synth.normalish(seed, IMG * IMG, 30 + b).reshape(IMG, IMG)
        ys, xs, amp = _blobs(seed, 7, 40 + b)
        for y, x, a in zip(ys, xs, amp):
            e += a * np.exp(-((rows - y) ** 2 + (cols - x) ** 2) / (2 * 4.0 ** 2))
        endp[b, 0] = e.astype(np.float32)
    out['semantic_seg'] = sem
    out['endp_est'] = endp
    return out


def expand_rows(rows144):
    """[144,1152] confidence at image rows 3::8 -> full [1152,1152] map (zeros elsewhere)."""
    full = np.zeros((IMG, IMG), dtype=np.float32)
    full[3::8, :] = rows144
    return full


def endp_map(pts):
    m = np.zeros((IMG, IMG), dtype=np.float32)
    for (h, w) in np.asarray(pts).reshape(-1, 2):
        m[int(h), int(w)] = 1.0
    return m


def postproc_case(i):
    """Hand-built / random decode results for the polyline assembly golden (G6).

    Returns dict(prop_conf1 [72] f32, prop_v_ext [72,144] u8, cls_offset [72,144] f64,
                 bi_seg_rows [144,1152] f32, endp_pts [K,2] int)."""
    seed = 6000 + i
    u = synth.uniform(seed, 4096, 0)
    H = 144
    hs = np.arange(H, dtype=np.float64)
    lanes = []   # (col_px(h) array, h0, h1, semantic pattern array)
    if i == 1:
        n_lane = 0
    elif i == 2:                                    # two crossing lanes
        lanes.append((300 + 3.0 * hs, 0, 143, np.ones(H)))
        lanes.append((700 - 2.6 * hs, 0, 143, np.full(H, 2.0)))
        n_lane = 0
    elif i == 3:                                    # a lane with a >24-row gap and one with a short gap
        sem = np.ones(H)
        lanes.append((400 + 0.5 * hs, 0, 143, sem))
        lanes.append((800 - 0.3 * hs, 0, 143, np.full(H, 2.0)))
        n_lane = 0
    elif i == 4:                                    # close parallel lanes (< 10 px and ~12 px apart)
        lanes.append((500 + 0.2 * hs, 0, 143, np.ones(H)))
        lanes.append((507 + 0.2 * hs, 10, 130, np.ones(H)))
        lanes.append((900 + 0.1 * hs, 0, 143, np.full(H, 2.0)))
        lanes.append((912 + 0.1 * hs, 0, 143, np.full(H, 2.0)))
        n_lane = 0
    else:
        n_lane = 4 if i == 0 else 1 + int(u[0] * 6)
    for k in range(n_lane):
        c0 = 80 + (IMG - 160) * (k + 0.3 + 0.4 * u[10 + k]) / max(n_lane, 1)
        slope = 1.6 * (u[20 + k] - 0.5)
        curve = 0.006 * (u[30 + k] - 0.5)
        h0 = int(u[40 + k] * 30) if u[50 + k] < 0.5 else 0
        h1 = 143 - (int(u[60 + k] * 30) if u[70 + k] < 0.5 else 0)
        col = c0 + slope * hs + curve * hs * hs
        kind = int(u[80 + k] * 3)
        if kind == 0:
            sem = np.ones(H)
        elif kind == 1:
            sem = np.full(H, 2.0)
        else:                                       # solid -> dashed change in the middle
            sem = np.where(hs < 40 + 60 * u[90 + k], 1.0, 2.0)
        lanes.append((col, h0, h1, sem))
    noise = synth.normalish(seed, 72 * H, 1).reshape(72, H)
    rnd = synth.uniform(seed, 72 * H, 2).reshape(72, H)
    rnd2 = synth.uniform(seed, 72 * H, 3).reshape(72, H)
    rndp = synth.uniform(seed, 72 * 8, 4).reshape(72, 8)
    ext = np.zeros((72, H), dtype=np.uint8)
    conf1 = (0.02 + 0.2 * rndp[:, 0]).astype(np.float64)
    # background: argmax bin + offset + proposal shift, anywhere in the proposal's field
    off = (np.floor(rnd * 10) + 0.3 * noise) + (2 * np.arange(72)[:, None] - 4)
    for li, (col, h0, h1, sem) in enumerate(lanes):
        g = col / 8.0
        for p in range(72):
            inside = (g >= 2 * p - 3) & (g < 2 * p + 5) & (hs >= h0) & (hs <= h1) & (col > 8) & (col < IMG - 8)
            if inside.sum() < 3 or rndp[p, 1 + (li % 6)] < 0.25:      # proposal did not fire for this lane
                continue
            conf1[p] = max(conf1[p], 0.55 + 0.44 * rndp[p, 7])
            rows = np.nonzero(inside & (rnd2[p] > 0.12))[0]           # 12 % vertex dropout
            if i == 3 and li == 0:
                rows = rows[(rows < 50) | (rows > 80)]                # 30-row gap
            if i == 3 and li == 1:
                rows = rows[(rows < 60) | (rows > 70)]                # 10-row gap
            flip = rnd[p, rows] < 0.06                                # 6 % wrong semantics
            s = sem[rows].copy()
            s[flip] = 3 - s[flip]
            ext[p, rows] = s.astype(np.uint8)
            off[p, rows] = g[rows] + 0.04 * noise[p, rows]
    cols = np.arange(IMG, dtype=np.float64)[None, :]
    conf = 0.05 + 0.1 * synth.uniform(seed, H * IMG, 5).reshape(H, IMG)
    for (col, h0, h1, sem) in lanes:
        on = ((hs >= h0) & (hs <= h1))[:, None]
        conf = conf + on * 0.8 * np.exp(-((cols - col[:, None]) ** 2) / (2 * 3.0 ** 2))
    pts = []
    for li, (col, h0, h1, sem) in enumerate(lanes):
        for hh in (h0, h1):
            if 3 <= hh <= 140 and 30 < col[hh] < IMG - 30 and u[200 + 2 * li + (hh == h1)] < 0.8:
                pts.append((int(8 * hh + 3 + 3 * (u[300 + li] - 0.5)), int(col[hh] + 4 * (u[320 + li] - 0.5))))
        ch = np.nonzero(np.diff(sem) != 0)[0]
        for c in ch:
            pts.append((int(8 * c + 5), int(col[c])))
        if u[340 + li] < 0.4:                                          # spurious endpoint in mid-line
            m = (h0 + h1) // 2
            pts.append((int(8 * m + 3), int(col[m]) + 2))
    for k in range(int(u[400] * 4)):                                   # endpoints far from any line
        pts.append((int(30 + u[410 + k] * 1090), int(30 + u[420 + k] * 1090)))
    if not pts:
        pts.append((600, 600))
    pts = np.unique(np.array(pts, dtype=np.int64).reshape(-1, 2), axis=0)
    pts = pts[(pts[:, 0] >= 20) & (pts[:, 0] < IMG - 20) & (pts[:, 1] >= 20) & (pts[:, 1] < IMG - 20)]
    return {'prop_conf1': conf1.astype(np.float32), 'prop_v_ext': ext, 'cls_offset': off.astype(np.float64),
            'bi_seg_rows': conf.astype(np.float32), 'endp_pts': pts}


# ---------------------------------------------------------------------------------------------- config 5 (LiDAR encoder)
def lidar_tail_input(seed, B=2, C=128, H=13, W=12):
    """Backbone-output-like tensor: post-ReLU (>= 0), ~60 % inactive sites (exact zeros)."""
    from lanemapping_amd import synth
    n = B * C * H * W
    v = np.abs(synth.normalish(seed, n)).astype(np.float32).reshape(B, C, H, W)
    act = (synth.uniform(seed + 1, B * H * W) > 0.6).reshape(B, 1, H, W)
    return v * act


def small_lidar_cfg(Xn=24, grid=96, sparse_hw=100, max_voxels=100000, max_points=10):
    """Config-5 pcencoder scaled down 6x in H/W (same layer stack) so that the dense oracle runs in seconds."""
    from lanemapping_amd.config import ConfigDict
    return ConfigDict(gt_downsample_ratio=8, pcencoder=dict(
        type='LidarEncoder', Xn=Xn, Yn=Xn, out_channels=64, lidar_encoder=dict(
            voxelize=dict(point_cloud_range=[-15., -25., -2., 15., 25., 2.], max_num_points=max_points,
                          grid_shape=[grid, grid, 10], max_voxels=max_voxels),
            backnone=dict(type='SparseEncoder', in_channels=4, sparse_shape=[21, sparse_hw, sparse_hw], output_channels=128,
                          order=('conv', 'norm', 'act'),
                          encoder_channels=((16, 16, 32), (32, 32, 64), (64, 64, 128), (128, 128)),
                          encoder_paddings=([0, 0, 1], [0, 0, 1], [0, 0, [1, 1, 0]], [0, 0]), block_type='basicblock'))))


# ---------------------------------------------------------------------------------------------- f1 (BEV polylines -> LAS frame)
def img2pc_case(seed, n_line=6, max_len=40, size=384):
    """(params, img_seqs [L,V,2], lens, tile u8 HWC) with empty regions under some vertices (big enough that the fill of one
    vertex feeds the window of the next), padding slots at (0, 0) and a non-unit quaternion."""
    from lanemapping_amd import synth
    tile = synth.bev_tile_u8(seed, size).copy()
    u = synth.uniform(seed, 64, 77)
    tile[int(60 + 40 * u[0]):int(150 + 40 * u[1]), int(40 + 30 * u[2]):int(200 + 60 * u[3]), :] = 0     # large hole
    tile[int(250 + 20 * u[4]):int(270 + 30 * u[5]), :, :] = 0                                              # empty band
    tile[300:303, 100:103, :] = [[[1, 0, 0]] * 3] * 3                                                      # channel sum == 1: "empty"
    lens = [int(2 + (max_len - 2) * u[8 + l]) for l in range(n_line)]
    lens[1] = 1                                                                                            # degenerate fit (q == 0)
    lens[2] = max_len
    seqs = np.zeros((n_line, max_len, 2))
    for l in range(n_line):
        r0 = 3 + 8 * int(4 * u[20 + l])
        rows = r0 + 8 * np.arange(lens[l])
        cols = 30 + (size - 60) * u[30 + l] + 0.15 * (rows - r0) * (u[40 + l] - 0.5) + synth.uniform(seed + l, lens[l], 78) * 3
        seqs[l, :lens[l], 0] = np.minimum(rows, size - 1)
        seqs[l, :lens[l], 1] = np.clip(cols, 1, size - 1)
    seqs[3, 2] = (301., 101.)                                                                              # lands on the sum == 1 pixel
    params = {'img_reso': [0.05, 0.0499], 'bev_img_offset': [-12.5 + u[50], 7.25 - u[51]], 'ele_reso': 0.0213,
              'local_min_ele': -3.4 + u[52], 'las_read_offset': [351200.0, 3433000.0, 12.0],
              'las_rotation_trans_quan': [18.25 + u[53], -7.5, 1.125, 0.92 + 0.1 * u[54], 0.013, -0.021, 0.38 + 0.1 * u[55]]}
    return params, seqs, lens, tile


def write_img2pc_files(d, seed):
    """Inputs of the per-tile driver: t.json (2-D polylines, reference writer format), t.png, t.txt (parameter file)."""
    import json
    from PIL import Image
    params, seqs, lens, tile = img2pc_case(seed)
    recs = [{'seq_len': int(n), 'seq': [[float(r), float(c), 1.0] for r, c in seqs[l, :n]], 'init_vertex': seqs[l, 0].tolist(),
             'end_vertex': seqs[l, n - 1].tolist()} for l, n in enumerate(lens)]
    with open(f'{d}/t.json', 'w') as f:
        json.dump(recs, f, indent=4)
    Image.fromarray(tile).save(f'{d}/t.png')
    sp = lambda v: ' '.join(repr(float(x)) for x in v)
    with open(f'{d}/t.txt', 'w') as f:
        f.write('\n'.join(['coor_las_path', 'synthetic.las', 'las_read_offset', sp(params['las_read_offset']),
                           'las_rotation_trans_quan', sp(params['las_rotation_trans_quan']), 'bev_img_offset',
                           sp(params['bev_img_offset']), 'img_reso', sp(params['img_reso']), 'local_min_ele',
                           repr(float(params['local_min_ele'])), 'ele_reso', repr(float(params['ele_reso'])), '']))


# ---------------------------------------------------------------------------------------------- f2 (cross-tile merge)
def merge_case_files(d, seed=701, n_tiles=5):
    """Writes tile_00.json .. (3-D polylines in the map frame, reference writer format) for a gently curving road seen by
    overlapping tiles: 3 lanes, lane 1 is digitised backwards in tile 2, lane 2 is missing from tile 3 (retired, then
    restarted), a 2-vertex stub appears in tile 1, a 4th lane starts in tile 2.  Returns the file list (unsorted)."""
    import json
    import os
    from lanemapping_amd import synth
    u = synth.uniform(seed, 4096, 91)
    files = []
    k = 0
    for t in range(n_tiles):
        x0, x1 = 20.0 * t, 20.0 * t + 31.0
        recs = []
        lanes = [0, 1, 2] + ([3] if t >= 2 else [])
        for lane in lanes:
            if lane == 2 and t == 3:
                continue
            xs = np.arange(x0 + 0.3 * lane, x1, 0.4)
            noise = (u[k:k + len(xs)] - 0.5) * 0.06
            k += len(xs)
            ys = 3.5 * lane + 0.002 * (xs - 40.0) ** 2 + noise
            zs = 12.0 + 0.01 * xs + (u[k:k + len(xs)] - 0.5) * 0.02
            k += len(xs)
            seq = np.stack([xs, ys, zs], axis=1)
            if lane == 1 and t == 2:
                seq = seq[::-1].copy()
            recs.append({'seq_len': int(len(seq)), 'seq': seq.tolist(), 'init_vertex': seq[0].tolist(), 'end_vertex': seq[-1].tolist()})
        if t == 1:
            stub = np.array([[x0 + 5.0, -6.0, 12.0], [x0 + 5.4, -6.0, 12.0]])
            recs.append({'seq_len': 2, 'seq': stub.tolist(), 'init_vertex': stub[0].tolist(), 'end_vertex': stub[1].tolist()})
        path = os.path.join(d, 'tile_%02d.json' % t)
        with open(path, 'w') as f:
            json.dump(recs, f, indent=4)
        files.append(path)
    return files[::-1]


# ---------------------------------------------------------------------------------------------- f3 (metrics)
def metric_case(seed):
    """(coor_label [12,144], cls_coors [72,144] with -1 = no vertex, endp_gt [1152,1152], endp_pred [1152,1152])."""
    from lanemapping_amd import synth
    u = synth.uniform(seed, 20000, 55)
    label = np.full((12, 144), -1.0)
    pred = np.full((72, 144), -1.0)
    rows = np.arange(144)
    for l in range(int(3 + 6 * u[0])):
        lo, hi = int(40 * u[10 + l]), 144 - int(40 * u[30 + l])
        col = 80 + 1000 * u[50 + l] + (rows - 72) * (u[70 + l] - 0.5)
        label[l, lo:hi] = np.clip(col[lo:hi], 1, 1151)
        if u[90 + l] > 0.2:                                                  # predicted, shifted by up to +-20 px, partly missing
            shift = 40 * (u[110 + l] - 0.5)
            plo, phi = lo + int(10 * u[130 + l]), hi - int(10 * u[150 + l])
            pred[4 + 5 * l, plo:phi] = np.clip(col[plo:phi] + shift + 6 * (u[200 + l * 144:200 + l * 144 + 144][plo:phi] - 0.5), 1, 1151)
    pred[70, 20:60] = 600.0 + 3 * (rows[20:60] % 3)                          # a false-positive lane
    egt = np.zeros((1152, 1152), np.float32)
    epr = np.zeros((1152, 1152), np.float32)
    for k in range(int(2 + 10 * u[3])):
        r, c = int(20 + 1100 * u[5000 + k]), int(20 + 1100 * u[5100 + k])
        egt[r, c] = 1.0
        if u[5200 + k] > 0.3:
            epr[min(1151, r + int(24 * (u[5300 + k] - 0.5))), min(1151, c + int(24 * (u[5400 + k] - 0.5)))] = 1.0
    epr[5, 5] = 1.0
    return label, pred, egt, epr


def label_json_text(seed):
    """A seeded annotation file in the schema data/convert_data.py reads: 5 boundary instances of different lengths, vertices with
    a third (ignored) component, integer semantics / instance ids."""
    from lanemapping_amd import synth
    u = synth.uniform(seed, 4000, 77)
    areas = []
    for k in range(5):
        n = 3 + int(40 * u[k])
        xs = 50 + 1000 * u[10 + k] + np.cumsum(8 * (u[100 + 60 * k:100 + 60 * k + n] - 0.5))
        ys = 20 + np.arange(n) * (1100.0 / n)
        seq = [[float(round(x, 3)), float(round(y, 3)), int(k)] for x, y in zip(xs, ys)]
        areas.append({'seq': seq, 'init_vertex': seq[0][0:2], 'end_vertex': seq[-1][0:2], 'semantic': int(1 + (k % 2)), 'instance': int(k + 1)})
    return json.dumps(areas)


# ---------------------------------------------------------------------------------------------- a12 (dataset contract) / f3 (evaluation loop)
def write_png(path, arr):
    """uint8 [H,W] / [H,W,3] -> 8-bit non-interlaced PNG (filter 0 on every scanline), zlib only."""
    import struct
    import zlib
    a = np.ascontiguousarray(arr, dtype=np.uint8)
    if a.ndim == 2:
        a = a[:, :, None]
    h, w, c = a.shape
    ctype = {1: 0, 3: 2, 4: 6}[c]
    raw = np.concatenate([np.zeros((h, 1), np.uint8), a.reshape(h, w * c)], axis=1).tobytes()

    def chunk(tag, data):
        return struct.pack('>I', len(data)) + tag + data + struct.pack('>I', zlib.crc32(tag + data) & 0xffffffff)
    with open(path, 'wb') as f:
        f.write(b'\x89PNG\r\n\x1a\n' + chunk(b'IHDR', struct.pack('>IIBBBBB', w, h, 8, ctype, 0, 0, 0))
                + chunk(b'IDAT', zlib.compress(raw, 1)) + chunk(b'IEND', b''))


def dataset_stems(n):
    """Tile stems in the 'YYMMDD_NNNN' pattern of the WHU-Lane files; every third one carries a suffix (names are cut to [0:11])."""
    return ['%06d_%04d%s' % (190712 + (i * 7) % 5, 519 + 37 * i, '_b' if i % 3 == 2 else '') for i in range(n)]


def label_case(seed, size=1152):
    """Seeded label set of one tile: (instance u8 [size,size] with 1-based lane ids, 0 = background, one id above number_lanes;
    semantic u8 (128 solid / 255 dashed, plus stray pixels outside every instance); endp u8 (255 at line ends); orient u8; areas =
    the sparse_seq JSON list).  Lines 2 and 3 form a connected pair (3 starts one row below the end of 2, one column off), line 1
    has two pixels on some rows."""
    u = synth.uniform(seed, 4000, 91)
    inst = np.zeros((size, size), np.uint8)
    sem = np.zeros((size, size), np.uint8)
    endp = np.zeros((size, size), np.uint8)
    ori = np.zeros((size, size), np.uint8)
    areas = []
    n = 5 + int(3 * u[0])
    spans = []
    for k in range(n):
        lo = 8 + int(300 * u[10 + k])
        hi = size - 8 - int(300 * u[30 + k])
        c0 = 80 + (size - 160) * (k + 0.5 * u[50 + k]) / n
        slope = 0.12 * (u[70 + k] - 0.5)
        spans.append([lo, hi, c0, slope])
    if n >= 4:                                     # lines 2 and 3: a connected pair
        spans[2][1] = size // 2
        lo3 = spans[2][1] + 1
        spans[3][0] = lo3
        spans[3][2] = spans[2][2] + spans[2][3] * (spans[2][1] - size / 2) - spans[3][3] * (lo3 - size / 2) + 1.0
    for k, (lo, hi, c0, slope) in enumerate(spans):
        ident = k + 1 if k != n - 1 else 14        # the last line carries an id above number_lanes = 12: it is dropped
        rows = np.arange(lo, hi + 1)
        cols = np.clip(np.rint(c0 + slope * (rows - size / 2)).astype(int), 2, size - 3)
        semantic = 1 + (k % 2)
        inst[rows, cols] = ident
        sem[rows, cols] = 128 if semantic == 1 else 255
        if k == 1:
            inst[rows[::3], cols[::3] + 1] = ident
            sem[rows[::3], cols[::3] + 1] = 128 if semantic == 1 else 255
        ori[rows, cols] = 1 + (k % 10)
        endp[rows[0], cols[0]] = 255
        endp[rows[-1], cols[-1]] = 255
        step = max(1, len(rows) // 12)
        seq = [[int(r), int(c)] for r, c in zip(rows[::step], cols[::step])]
        areas.append({'seq': seq, 'init_vertex': [int(rows[0]), int(cols[0])], 'end_vertex': [int(rows[-1]), int(cols[-1])],
                      'semantic': int(semantic), 'instance': int(ident)})
    sem[5, 5:40] = 255                             # semantic pixels outside every instance: cleared by the loader
    return inst, sem, endp, ori, areas


def write_dataset(root, n_tiles=5, seed=1801, size=1152, split_file='data_split-shuffle.json', label_dir='labels', tile_seeds=None):
    """A synthetic <data_root> in the reference's layout: split file, cropped_tiff/<stem>.png, <label_dir>/sparse_*/<stem>.{png,json}.
    Returns the stems.  Split lists: test = all tiles, pretrain = all but the last, valid = first two, single = first, train = last."""
    stems = dataset_stems(n_tiles)
    for sub in ('cropped_tiff', label_dir + '/sparse_seq', label_dir + '/sparse_semantic', label_dir + '/sparse_instance',
                label_dir + '/sparse_orient', label_dir + '/sparse_endp'):
        os.makedirs(os.path.join(root, sub), exist_ok=True)
    for i, s in enumerate(stems):
        ts = (tile_seeds[i] if tile_seeds is not None else seed + i)
        write_png(os.path.join(root, 'cropped_tiff', s + '.png'), synth.bev_tile_u8(ts, size))
        inst, sem, endp, ori, areas = label_case(seed + 100 + i, size)
        write_png(os.path.join(root, label_dir, 'sparse_instance', s + '.png'), inst)
        write_png(os.path.join(root, label_dir, 'sparse_semantic', s + '.png'), sem)
        write_png(os.path.join(root, label_dir, 'sparse_endp', s + '.png'), endp)
        write_png(os.path.join(root, label_dir, 'sparse_orient', s + '.png'), ori)
        with open(os.path.join(root, label_dir, 'sparse_seq', s + '.json'), 'w') as f:
            json.dump(areas, f)
    with open(os.path.join(root, split_file), 'w') as f:
        json.dump({'train': stems[-1:], 'test': stems, 'valid': stems[:2], 'single': stems[:1], 'pretrain': stems[:-1]}, f)
    return stems
