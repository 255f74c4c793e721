"""Dataset contract of the inference / evaluation harness: which files a split names, and the ground truth the
test loop scores against.  Listing + evaluation GT only; the training-target construction of the reference is out of scope.

  load_datadir(data_root, data_split_file, mode, ...)  <- baseline/datasets/laserlane_proposals.py:498-536 (laserlane.py:125-162:
        same with the split file fixed to 'data_split-shuffle.json')
        <data_root>/<data_split_file> is a JSON object with the lists 'train' / 'test' / 'valid' / 'single' / 'pretrain' of tile stems;
        mode 'single' | 'valid' (first 150) | 'test' | 'all' / 'infer_only' (the 'pretrain' list) | anything else -> 'train'.
        Images are <data_root>/cropped_tiff/<stem>.png, labels <data_root>/<label_dir>/sparse_{seq,semantic,instance,orient,endp}/<stem>.*
        (:40-47).  The reference shuffles the 'test' list with the global `random` that Runner.__init__ seeded with cfg.seed
        (runner.py:69-71): `shuffle_seed` reproduces that order with a private random.Random(seed) - the same Mersenne
        Twister stream - and None keeps the file's order.  The Runner itself walks the tiles SORTED by stem (per-tile results do
        not depend on the order; a static rank shard needs a canonical one).
  split_entries(split_cfg, cfg)                        <- LaserLaneProposal.__init__ :38-65, LaserLane.__init__ (laserlane.py:33-58),
        LaserLaneProposalEgo.__init__ (laserlane_proposals_ego.py:39-72): the dataset `type` picks label directory / split file; the
        mode assertion of each class is kept (mode 'val', which the published configs carry for their `val` split, fails it: C12).
  load_eval_gt(entry, cfg, merge_connect_lines)        <- the part of format_gt_column_proposal (:102-252) the test loop reads
        (runner.py:736-787): 'lc_coor_raw' [number_lanes,144] (GT column px on the rows 3::8, -1 = none; :267-387 + :414-496),
        'mask' (semantic label image, 128 -> 1, 255 -> 2, zero outside the kept instances; load_label_image :588-615, :113-116) and
        'endp_map' (endpoint label image / 255).  Label PNGs are read with the library's own PNG reader.
"""
import json
import os
import os.path as osp
import random

import numpy as np

IMG_DIR = 'cropped_tiff'
_MODES = {
    'LaserLaneProposal': {"train", "valid", "test", "single", "all", "infer_only"},     # laserlane_proposals.py:39
    'LaserLaneProposalEgo': {"train", "valid", "test", "single", "all"},                # laserlane_proposals_ego.py:40
    'LaserLane': {"train", "valid", "test", "single", "all"},                           # laserlane.py:34
}
_LABEL_DIR = {'LaserLaneProposal': 'labels', 'LaserLaneProposalEgo': 'labels_inside_lidar_range', 'LaserLane': 'labels'}


def load_datadir(data_root, data_split_file, mode, label_dir='labels', shuffle_seed=None, infer_only_is_all=True):
    with open(osp.join(data_root, data_split_file), 'r') as jf:
        split = json.load(jf)
    lists = {k: list(split[k]) for k in ('train', 'test', 'valid', 'single', 'pretrain')}     # (a missing key is a KeyError there too)
    if mode == 'single':
        stems = lists['single']
    elif mode == 'valid':
        stems = lists['valid'][:150]
    elif mode == 'test':
        stems = lists['test']
        if shuffle_seed is not None:
            random.Random(shuffle_seed).shuffle(stems)
    elif mode == 'all' or (mode == 'infer_only' and infer_only_is_all):
        stems = lists['pretrain']
    else:
        stems = lists['train']
    lab = osp.join(data_root, label_dir)
    return [{'stem': s,
             'image': osp.join(data_root, IMG_DIR, s + '.png'),
             'seq': osp.join(lab, 'sparse_seq', s + '.json'),
             'mask': osp.join(lab, 'sparse_semantic', s + '.png'),
             'instance': osp.join(lab, 'sparse_instance', s + '.png'),
             'ori': osp.join(lab, 'sparse_orient', s + '.png'),
             'endp': osp.join(lab, 'sparse_endp', s + '.png')} for s in stems]


def split_entries(split_cfg, cfg=None, shuffle_seed=None):
    """A config's split dictionary ({type, data_root, [data_split_file,] mode}: cfg.dataset.test / .infer_only / ...) -> entries."""
    kind = split_cfg['type']
    if kind not in _MODES:
        raise KeyError(f'{kind} is not in the dataset registry')               # utils/registry.py:70-72 wording
    mode = split_cfg.get('mode', 'valid')
    assert mode in _MODES[kind], f'dataset mode {mode!r} is not one of {sorted(_MODES[kind])}'
    split_file = 'data_split-shuffle.json' if kind == 'LaserLane' else split_cfg['data_split_file']
    return load_datadir(split_cfg['data_root'], split_file, mode, label_dir=_LABEL_DIR[kind], shuffle_seed=shuffle_seed,
                        infer_only_is_all=(kind == 'LaserLaneProposal'))


# ------------------------------------------------------------------------------------------------ evaluation ground truth
def _label_seq_points(seq_path, number_lanes):
    """init / end vertex and semantic of the first `number_lanes` annotated lines, zero padded (:119-131)."""
    with open(seq_path) as f:
        data = json.load(f)
    initp = np.zeros((number_lanes, 2), dtype=np.float64)
    endp = np.zeros((number_lanes, 2), dtype=np.float64)
    sem = np.zeros((number_lanes,), dtype=np.float64)
    n = min(len(data), number_lanes)
    if n:
        initp[:n] = np.array([a['init_vertex'] for a in data[:n]], dtype=np.float64)[:, 0:2]
        endp[:n] = np.array([a['end_vertex'] for a in data[:n]], dtype=np.float64)[:, 0:2]
        sem[:n] = np.array([a['semantic'] for a in data[:n]], dtype=np.float64)
    return initp, endp, sem


def lane_columns_from_instance(label_raw, number_lanes, row_size=144, ratio=8):
    """[H,W] instance label (lane id 0..n-1, 255 = background) -> [n,row_size] f32: the column of lane c on image row 3 + 8 h divided
    by `ratio` (when a row holds several pixels of a lane the right-most one stays: the reference's indexed assignment visits them in
    row-major order), 0 -> -1 (:425-443)."""
    H = label_raw.shape[0]
    cls_maps = np.zeros((number_lanes, row_size), dtype=np.float32)
    for c in range(number_lanes):
        r, col = np.nonzero(label_raw == c)
        raw = np.zeros((H,), dtype=np.float32)
        raw[r] = col.astype(np.float32) / np.float32(ratio)
        cls_maps[c] = raw[3:H:8][:row_size]
    cls_maps[cls_maps == 0.] = -1.
    return cls_maps


def load_eval_gt(entry, cfg, merge_connect_lines=True):
    from .png_io import read_png
    L = int(cfg.number_lanes)
    ratio = int(cfg.get('gt_downsample_ratio', 8))
    row_size = int(cfg.heads.row_size) if 'heads' in cfg and 'row_size' in cfg.heads else 144
    mask = read_png(entry['mask']).copy()
    mask[mask == 128] = 1
    mask[mask == 255] = 2
    inst = read_png(entry['instance'])
    endp_map = read_png(entry['endp']).astype(np.float32) / np.float32(255.)
    inst = np.where(inst > L, 0, inst)
    mask = np.where(inst == 0, 0, mask)
    label_raw = np.where(inst == 0, 255, inst - 1)
    if cfg.get('flip_label', False):
        label_raw = label_raw[::-1, ::-1]
    initp, endp, sem = _label_seq_points(entry['seq'], L)
    cls = lane_columns_from_instance(label_raw, L, row_size, ratio)
    exist = np.where(cls > 0., sem[:, None].astype(np.float32), np.float32(0.))
    if merge_connect_lines:                     # a line that starts where another one ends is appended to it (:330-363)
        for a in range(L):
            end1 = endp[a]
            if end1[0] > 0 and end1[1] > 0:
                for b in range(L):
                    if b == a:
                        continue
                    s2 = initp[b]
                    if s2[0] > 0 and s2[1] > 0 and abs(end1[0] - s2[0]) < 2 and abs(end1[1] - s2[1]) < 2:
                        rows = np.nonzero(exist[b] > 0)[0]
                        exist[a, rows] = exist[b, rows]
                        cls[a, rows] = cls[b, rows]
                        exist[b, rows] = 0
                        cls[b, rows] = -1
                        initp[b] = 0
                        endp[b] = 0
    lc = cls.copy()
    lc[lc > -1.] *= np.float32(ratio)
    return {'lc_coor_raw': lc, 'mask': mask, 'endp_map': endp_map, 'label_raw': label_raw}


def klane_coor_label(label_raw, num_cls, row_size=144):
    """config 4's `coor_label` (row_shared_not_reduc_ref.py:486, :676-710 with downsample=False): column px per lane and row, no
    merging."""
    return lane_columns_from_instance(label_raw, num_cls, row_size, 1)
