"""Oracle (test infrastructure, never imported by the product): cross-tile merge of the 3-D polylines, SURVEY §8f row f2.
The product is host C++ behind the C-ABI (lanemapping_amd/csrc/merge_lines.cpp, lm_merge_*); this numpy restatement is what it
is checked against, and is itself pinned by golden G13 (output of the imported reference).

Restates baseline/utils/merge_lines.py (`merge_lines` :166-291, `merge_2_seqs` :67-104, `merge_2_reversed_seqs` :106-132,
`downsample_seqs` :133-153, distance / orientation helpers :17-65, :157-164): same function names, argument meaning and
results — including the behaviours a tidy rewrite would lose and that decide which lines survive:
  * tiles are visited in sorted file-name order and a line is matched through its FIRST vertex only;
  * the retirement loop removes entries from the lists it is iterating over, so the entry after every retired line is
    skipped for that round (:270-281);
  * `load_lane_seq` returns no sequences for a file with fewer than two lines (io_utils.py:116-117).
This is the sequential host step after the per-tile back-projection (`coor_img2pc`), O(lines^2) per tile and tiny next
to the GPU work; it is plain numpy on purpose so that every comparison (`<`, argmin, eig) is the reference's own.
"""
import numpy as np

from lanemapping_amd.io_utils import load_lane_seq

EPS = 1e-6


def calculate_principal_easy(seq):
    """Unit vector first -> last vertex in the xy plane (:58-64)."""
    d = seq[-1, :] - seq[0, :]
    d[2] = 0
    return d / (np.sqrt(np.sum(np.square(d))) + EPS)


def calculate_principal_strict(seq):
    """Dominant eigenvector of the xy scatter matrix (:46-56)."""
    seq = np.array(seq)
    assert len(seq) >= 2
    c = seq - np.mean(seq, axis=0)
    c[:, 2] = 0
    vals, vecs = np.linalg.eig(np.dot(c.transpose(), c))
    return vecs[:, np.argsort(vals)[-1]]


def calculate_neatest_dist_id(pt, seq, proj=False):
    """(distance, index) of the vertex of `seq` nearest to `pt` in xy; with `proj` the distance is measured perpendicular
    to the sequence's first->last direction through that vertex (:17-31)."""
    d2 = np.square(seq[:, 0] - pt[0]) + np.square(seq[:, 1] - pt[1])
    k = np.argmin(d2)
    dist = np.sqrt(d2[k])
    if proj:
        perp = np.cross(calculate_principal_easy(seq), pt - seq[k, :])
        dist = np.sqrt(perp[0] * perp[0] + perp[1] * perp[1] + perp[2] * perp[2])
    return dist, k


def calculate_average_distance_in_overlap_area(seq_query, seq_key):
    """Mean / max nearest-vertex distance from every vertex of `seq_key` to `seq_query` (:36-44)."""
    total, worst = 0., 0.
    for v in seq_key:
        d, _ = calculate_neatest_dist_id(v, seq_query)
        total += d
        if d > worst:
            worst = d
    return total / seq_key.shape[0], worst


def _direction(seq_base):
    axis = calculate_principal_strict(seq_base)
    if np.dot(axis, calculate_principal_easy(seq_base)) < 0:
        axis *= -1
    return axis


def merge_2_seqs(seq_base, seq_new):
    """Weave `seq_new` (same heading) into the tail of `seq_base` by position along the base's principal axis; returns the
    merged sequence and the index in the base where the overlap starts (:67-104)."""
    axis = _direction(seq_base)
    base_t = [v.dot(axis) for v in seq_base]
    new_t = [v.dot(axis) for v in seq_new]
    base_over = np.where(base_t > new_t[0])
    new_over = np.where(new_t < base_t[-1])
    for j in new_over[0]:
        for i in base_over[0]:                       # (indices of the ORIGINAL base; the base grows while we insert)
            if new_t[j] < base_t[i]:
                seq_base = np.insert(seq_base, i, seq_new[j], axis=0)
                base_t = np.insert(base_t, i, new_t[j])
                break
    if len(new_over[0]) < 1:
        base_over = [[seq_base.shape[0]]]
        seq_base = np.insert(seq_base, seq_base.shape[0], seq_new, axis=0)
    else:
        seq_base = np.insert(seq_base, seq_base.shape[0], seq_new[(new_over[0][-1] + 1):, ], axis=0)
    return seq_base, base_over[0][0]


def merge_2_reversed_seqs(seq_base, seq_new):
    """`seq_new` runs against the base: append what lies beyond the base's end (walking it backwards), prepend what lies
    before its start (:106-132)."""
    axis = _direction(seq_base)
    base_t = [v.dot(axis) for v in seq_base]
    new_t = [v.dot(axis) for v in seq_new]
    ahead = np.where(new_t > base_t[-1])[0]
    behind = np.where(new_t < base_t[0])[0]
    for k in range(len(ahead)):
        j = ahead[-1 - k]
        seq_base = np.insert(seq_base, len(seq_base), seq_new[j], axis=0)
    for j in behind:
        seq_base = np.insert(seq_base, 0, seq_new[j], axis=0)
    return seq_base


def downsample_seqs(seq_base, dist_min=0.6):
    """Keep a vertex whenever more than `dist_min` metres (xy) accumulated since the last kept one (:133-153)."""
    nxt = seq_base.copy()
    nxt[:-1, :] = nxt[1:, :]
    step = nxt - seq_base
    step[:, 2] = 0
    step = np.sqrt(np.sum(np.square(step), axis=1))
    acc = 0.
    out = np.array([seq_base[0, :]])
    for i, d in enumerate(step):
        acc += d
        if acc > dist_min:
            out = np.concatenate((out, [seq_base[i, :]]), axis=0)
            acc = 0.
        elif i == (len(seq_base) - 1):
            if step[i - 1] < 0.05 or i == 0 or acc < 0.05:
                continue
            out = np.concatenate((out, [seq_base[i, :]]), axis=0)
    return out


def cal_local_orient(seq):
    """Heading from the last 5 vertices (:157-164)."""
    return calculate_principal_easy(seq[-5:, :]) if len(seq) > 5 else calculate_principal_easy(seq)


class _Active:
    """Parallel lists of the reference's bookkeeping (sequence, length, first/last point, overlap start, heading)."""

    def __init__(self):
        self.seq, self.length, self.init, self.end, self.roi, self.flag = [], [], [], [], [], []
        self.orient = np.zeros((0, 3))

    def append(self, seq, length, init, end, orient, flag=None):
        self.seq.append(seq)
        self.length.append(length)
        self.init.append(init)
        self.end.append(end)
        self.roi.append(0)
        self.orient = np.append(self.orient, [orient], axis=0)
        if flag is not None:
            self.flag.append(flag)

    def pop(self, i):
        seq = self.seq.pop(i)
        self.length.pop(i)
        self.init.pop(i)
        self.end.pop(i)
        self.roi.pop(i)
        self.flag.pop(i)
        self.orient = np.delete(self.orient, i, axis=0)
        return seq


def _load(path):
    seqs, lens, init, end = load_lane_seq(path, dim_coor=3)
    return [s[:lens[i]] for i, s in enumerate(seqs)], lens, init, end


def merge_lines(seq_filenames, verbose=False):
    """List of per-tile 3-D polyline JSON files -> list of merged [n, 3] arrays (:166-291)."""
    names = sorted(seq_filenames)
    done = []
    act = _Active()
    seqs, lens, init, end = _load(names[0])
    for i, s in enumerate(seqs):
        act.append(s, lens[i], init[i], end[i], cal_local_orient(s))
    for name in names[1:]:
        seqs, lens, init, end = _load(name)
        act.flag = [0] * len(act.seq)
        heads = np.zeros((len(seqs), 3))
        for i, s in enumerate(seqs):
            heads[i, :] = cal_local_orient(s)
        for t, cand in enumerate(seqs):
            best, best_d = -1, 10
            for a, base in enumerate(act.seq):
                d, _ = calculate_neatest_dist_id(init[t], base[int(act.roi[a]):, :], proj=True)
                if d < best_d:
                    best, best_d = a, d
            if best_d < 0.5:
                cos = heads[t].dot(act.orient[best])
                back_d, _ = calculate_neatest_dist_id(act.seq[best][-1, :], cand, proj=True)
                if verbose:
                    print('match', t, '->', best, 'dist', best_d, 'cos', cos)
                if back_d < 0.5 and cos > 0.7:                                    # same heading: weave into the tail
                    r = act.roi[best]
                    merged, start = merge_2_seqs(act.seq[best][r:, :], cand)
                    act.seq[best] = np.concatenate((act.seq[best][:r], merged), axis=0)
                    act.roi[best] += start
                    act.length[best] = len(act.seq[best])
                    act.end[best] = act.seq[best][-1, :]
                    act.flag[best] = 1
                    act.orient[best, :] = cal_local_orient(act.seq[best])
                    continue
                if back_d < 0.5 and cos < -0.7:                                   # opposite heading
                    act.seq[best] = merge_2_reversed_seqs(act.seq[best], cand)
                    act.length[best] = len(act.seq[best])
                    act.end[best] = act.seq[best][-1, :]
                    act.init[best] = act.seq[best][0, :]
                    act.flag[best] = 1
                    act.orient[best, :] = cal_local_orient(act.seq[best])
                    continue
            act.append(cand, lens[t], init[t], end[t], heads[t, :], flag=1)       # a new line starts here
        # retire the lines this tile did not touch.  The reference pops from the lists it enumerates, so the index keeps
        # advancing over a shrinking list: the entry that slides into a retired slot is not examined in this round.
        i = 0
        while i < len(act.flag):
            if act.flag[i] < 0.5:
                seq = act.pop(i)
                if len(seq) >= 3:
                    done.append(seq)
            i += 1
    for s in act.seq:
        if len(s) >= 3:
            done.append(s)
    return done
