"""CPU tests of the §8(f) evaluation / data-format rows: skeleton-based semantic F1 (f3) and the label-JSON schema (f4)."""
import json
import os

import numpy as np
import pytest
import scipy.ndimage as ndi

from lanemapping_amd import metric_utils as mu


def _components8(a):
    return ndi.label(a, structure=np.ones((3, 3)))[1]


def _holes(a):
    """Background components (4-connectivity) that do not touch the border = holes of the 8-connected object."""
    lab, n = ndi.label(np.pad(a == 0, 1, constant_values=True))
    return n - 1


def _random_shapes(seed, H=48, W=56):
    rng = np.random.RandomState(seed)
    img = np.zeros((H, W), np.uint8)
    for _ in range(rng.randint(2, 6)):                       # thick polylines
        r, c = rng.randint(4, H - 4), rng.randint(4, W - 4)
        for _ in range(rng.randint(10, 60)):
            t = rng.randint(1, 4)
            img[max(0, r - t):r + t, max(0, c - t):c + t] = 1
            r = int(np.clip(r + rng.randint(-2, 3), 2, H - 3))
            c = int(np.clip(c + rng.randint(-1, 4), 2, W - 3))
    if seed % 2:                                             # a ring (a hole must survive)
        yy, xx = np.mgrid[:H, :W]
        d = np.hypot(yy - H // 2, xx - W // 2)
        img[(d > 8) & (d < 13)] = 1
    return img


@pytest.mark.parametrize('seed', range(6))
def test_skeleton_cpp_vs_3d_oracle(seed):
    """lm_skeletonize_lee_2d (2-D tables) == the literal 3-D restatement (26-neighbourhoods, cubical-complex Euler characteristic,
    flood-fill simple-point test) on random thick polylines and rings."""
    from oracle import skeleton_ref
    img = _random_shapes(seed)
    got = mu.skeletonize_lee(img)
    want = skeleton_ref.skeletonize_lee_ref(img)
    assert np.array_equal(got, want)


@pytest.mark.parametrize('seed', range(4))
def test_skeleton_properties(seed):
    """What a thinning must keep: subset of the object, same 8-connected components, same holes, idempotent; and what it must
    achieve: no 2x2 block of skeleton pixels survives."""
    img = _random_shapes(10 + seed, 160, 200)
    sk = mu.skeletonize_lee(img)
    assert sk.dtype == np.uint8 and set(np.unique(sk)) <= {0, 1} and np.all(sk <= (img != 0))
    assert _components8(sk) == _components8(img) and _holes(sk) == _holes(img)
    assert np.array_equal(mu.skeletonize_lee(sk), sk)
    blocks = sk[:-1, :-1] & sk[1:, :-1] & sk[:-1, 1:] & sk[1:, 1:]
    assert not blocks.any()
    assert sk.sum() < 0.5 * img.sum()


def test_skeleton_known_answers():
    bar = np.zeros((15, 40), np.uint8)
    bar[4:11, 3:37] = 1                                      # a 7 x 34 bar thins to one horizontal line on its middle row
    sk = mu.skeletonize_lee(bar)
    rows = np.unique(np.nonzero(sk)[0])
    assert list(rows) == [7] and sk[7].sum() >= 26 and np.all(np.diff(np.nonzero(sk[7])[0]) == 1)
    assert mu.skeletonize_lee(np.zeros((5, 5))).sum() == 0
    one = np.zeros((5, 5), np.uint8)
    one[2, 2] = 1
    assert np.array_equal(mu.skeletonize_lee(one), one)      # an isolated pixel is not deletable
    line = np.zeros((9, 9), np.uint8)
    line[4, 1:8] = 1
    assert np.array_equal(mu.skeletonize_lee(line), line)    # already thin: end points are kept


def test_eval_metric_line_segmentor():
    """Counts of the reference's accounting (metric_utils.py:415-481) on hand-made maps: perfect match, shifted line inside / outside
    the buffer, per-class mode, and the empty-side branches."""
    H = W = 128
    gt = np.zeros((H, W), np.uint8)
    gt[10:110, 40] = 1
    gt[10:110, 90] = 2
    seg = np.zeros((H, W), np.uint8)
    seg[10:110, 38:43] = 1                                   # 5 px wide stripes around the GT columns
    seg[10:110, 88:93] = 2
    acc, rec, f, TP, n_seg, DG, n_gt = mu.eval_metric_line_segmentor(seg, gt, bi_seg=False, semantics=2, buff=10)
    assert (acc, rec, f) == (1.0, 1.0, 1.0) and n_gt == 200 and TP == n_seg and DG == n_gt
    a2 = mu.eval_metric_line_segmentor(seg, gt, bi_seg=True, buff=10)
    assert a2[:3] == (1.0, 1.0, 1.0) and a2[6] == 200
    far = np.zeros_like(seg)
    far[10:110, 58:63] = 1                                   # 20 px from class 1's GT: outside the buffer
    acc, rec, f, TP, n_seg, DG, n_gt = mu.eval_metric_line_segmentor(far, gt, bi_seg=False, semantics=2, buff=10)
    assert TP == 0 and DG == 0 and f == 0 and n_gt == 200 and n_seg > 80      # class 2: GT but no skeleton -> its GT pixels are missed
    swapped = np.where(seg == 1, 2, np.where(seg == 2, 1, 0))
    assert mu.eval_metric_line_segmentor(swapped, gt, bi_seg=False, semantics=2, buff=10)[2] == 0          # wrong classes
    assert mu.eval_metric_line_segmentor(swapped, gt, bi_seg=True, buff=10)[2] == 1.0                       # class-blind mode
    none = mu.eval_metric_line_segmentor(seg, np.zeros_like(gt), bi_seg=False, semantics=2)
    assert none[3] == 0 and none[4] > 150 and none[6] == 0 and none[2] == 0                                  # no GT: all false positives
    assert mu.eval_metric_line_segmentor(np.zeros_like(seg), np.zeros_like(gt))[:3] == (0., 0., 0)


def test_label_json_golden_g16(golden, tmp_path):
    """Label-JSON schema (data/convert_data.py:25-70): load_seq / save_seq vs the imported reference on a seeded annotation file."""
    import cases
    from lanemapping_amd import io_utils
    g = golden('g16_label_json.npz')
    src = tmp_path / 'label.json'
    src.write_text(cases.label_json_text(int(g['seed'])))
    seq, lens, sem, inst, init, end = io_utils.load_label_seq(str(src))
    assert np.array_equal(seq, g['seq']) and list(lens) == list(g['seq_lens'])
    assert list(sem) == list(g['semantic']) and list(inst) == list(g['instance'])
    assert np.array_equal(np.asarray(init, dtype=np.float64), g['init']) and np.array_equal(np.asarray(end, dtype=np.float64), g['end'])
    out = tmp_path / 'out.json'
    io_utils.save_label_seq(seq, lens, sem, inst, g['orient'], str(out))
    assert out.read_text() == str(g['saved_text'])
    back = io_utils.load_label_seq(str(out))
    assert np.array_equal(back[0], seq) and list(back[1]) == list(lens)


def test_line8_opencv_table():
    """cv2.line(thickness=1) as restated from OpenCV's LineIterator (csrc/postproc.cpp line8_opencv; oracle/postproc_ref.py _line8, written
    independently): a HAND-DERIVED table for the segments renew_semantic_map draws (dy = 8 between consecutive anchor rows).  Walk from the
    LEFT end point, err = major - 2 minor, a minor-axis move when err < 0 (err += 2 major - 2 minor), else err -= 2 minor:
      dx = 4 (y major, err 0 -8 0 -8 ...): x moves on steps 2, 4, 6, 8;   dx = -4: the same walk from the other (left) end, upwards;
      dx = 12 (x major, err -4 4 -12 -4 4 -12 ...): y moves on steps 1, 3, 4, 6, 7, 9, 10, 12.
    The textbook all-octant Bresenham used until round 4 takes the diagonal at err == 0 and fails this table (dx = 2, 4, 6)."""
    import ctypes as C
    from lanemapping_amd._lib import lib, check
    from oracle import postproc_ref
    table = {      # (dx, dy) -> x offset per row (y major) or (x, y) pairs (x major), relative to the FIRST end point (x0, y0)
        (0, 8): [0, 0, 0, 0, 0, 0, 0, 0, 0],
        (1, 8): [0, 0, 0, 0, 0, 1, 1, 1, 1],              # err 6 4 2 0 -2 | 12 ...: the one move on step 5
        (2, 8): [0, 0, 0, 1, 1, 1, 1, 2, 2],              # err 4 0 -4 8 4 0 -4 8: moves on steps 3, 7
        (4, 8): [0, 0, 1, 1, 2, 2, 3, 3, 4],
        (8, 8): [0, 1, 2, 3, 4, 5, 6, 7, 8],              # |dy| > dx is false: x major, err -8 each step -> every step diagonal
        (-4, 8): [0, -1, -1, -2, -2, -3, -3, -4, -4],     # walked from the left end (row 8) upwards: row 8 - k has offset -4 + [0,0,1,1,2,2,3,3,4][k]
        (-2, 8): [0, 0, -1, -1, -1, -1, -2, -2, -2],
    }
    xmajor = {(12, 8): [(0, 0), (1, 1), (2, 1), (3, 2), (4, 3), (5, 3), (6, 4), (7, 5), (8, 5), (9, 6), (10, 7), (11, 7), (12, 8)],
              (-12, 8): [(-12, 8), (-11, 7), (-10, 7), (-9, 6), (-8, 5), (-7, 5), (-6, 4), (-5, 3), (-4, 3), (-3, 2), (-2, 1), (-1, 1), (0, 0)]}
    x0, y0 = 500, 403
    for (dx, dy), want in list(table.items()) + list(xmajor.items()):
        pix = {(x0 + a, y0 + b) for a, b in want} if (dx, dy) in xmajor else {(x0 + a, y0 + r) for r, a in enumerate(want)}
        img = np.zeros((1152, 1152), np.uint8)
        check(lib().lm_line8(img.ctypes.data_as(C.c_void_p), x0, y0, x0 + dx, y0 + dy, 1))
        got = {(int(c), int(r)) for r, c in zip(*np.nonzero(img))}
        assert got == pix, ((dx, dy), sorted(got ^ pix))
        ref = np.zeros((1152, 1152))
        postproc_ref._line8(ref, x0, y0, x0 + dx, y0 + dy, 1)
        assert np.array_equal(ref != 0, img != 0), (dx, dy)
        back = np.zeros((1152, 1152), np.uint8)             # end-point order does not matter (leftToRight)
        check(lib().lm_line8(back.ctypes.data_as(C.c_void_p), x0 + dx, y0 + dy, x0, y0, 1))
        assert np.array_equal(back, img), (dx, dy)
    # every dx the assembly can produce between two rows, product vs oracle, plus clipping at the map border
    for dx in list(range(-40, 41)) + [-300, 300, 1151]:
        a, b = np.zeros((1152, 1152), np.uint8), np.zeros((1152, 1152))
        check(lib().lm_line8(a.ctypes.data_as(C.c_void_p), 10, 1139, 10 + dx, 1147, 2))
        postproc_ref._line8(b, 10, 1139, 10 + dx, 1147, 2)
        assert np.array_equal(a, b.astype(np.uint8)), dx


def _endpoints(sk):
    """Skeleton pixels with exactly one 8-neighbour."""
    nb = ndi.convolve(sk.astype(np.int32), np.ones((3, 3), np.int32), mode='constant') - sk
    return int(((sk == 1) & (nb == 1)).sum())


def test_skeleton_published_shapes():
    """The shapes thinning papers show (Lee, Kashyap & Chu 1994, figs. of rectangles, crosses and rings): what the published algorithm
    GUARANTEES on them - topology (components, holes), a one-pixel-wide curve, the number of free ends, medial position within one pixel -
    asserted on lm_skeletonize_lee_2d and on the literal 3-D restatement (oracle/skeleton_ref.py).  skimage is not importable in the build
    container, so the pixel-exact answer of skimage.morphology.skeletonize(method='lee') itself stays UNPINNED."""
    from oracle import skeleton_ref
    shapes = {}
    rect = np.zeros((20, 50), np.uint8); rect[6:12, 5:45] = 1                      # 6 x 40 rectangle (even thickness)
    shapes['rectangle'] = (rect, dict(comp=1, holes=0, ends=2))
    cross = np.zeros((41, 41), np.uint8); cross[17:24, 3:38] = 1; cross[3:38, 17:24] = 1      # 7-wide plus sign
    shapes['cross'] = (cross, dict(comp=1, holes=0, ends=4))
    yy, xx = np.mgrid[:61, :61]
    rr = np.hypot(yy - 30, xx - 30)
    ring = ((rr >= 14) & (rr <= 22)).astype(np.uint8)                                # annulus, 9 px thick
    shapes['ring'] = (ring, dict(comp=1, holes=1, ends=0))
    two = np.zeros((30, 60), np.uint8); two[4:9, 4:56] = 1; two[18:25, 10:50] = 1    # two separate bars
    shapes['two bars'] = (two, dict(comp=2, holes=0, ends=4))
    for name, (img, want) in shapes.items():
        sk = mu.skeletonize_lee(img)
        assert np.array_equal(sk, skeleton_ref.skeletonize_lee_ref(img)), name
        assert np.all(sk <= img) and _components8(sk) == want['comp'] and _holes(sk) == want['holes'], name
        assert _endpoints(sk) == want['ends'], (name, _endpoints(sk))
        assert not (sk[:-1, :-1] & sk[1:, :-1] & sk[:-1, 1:] & sk[1:, 1:]).any(), name          # one pixel wide
        assert np.array_equal(mu.skeletonize_lee(sk), sk), name                                    # idempotent
    r = mu.skeletonize_lee(shapes['rectangle'][0])
    rows = np.unique(np.nonzero(r)[0])
    assert set(rows) <= {8, 9} and r.sum() >= 40 - 6            # on one of the two middle rows, at most half the thickness shorter at each end
    ringsk = mu.skeletonize_lee(ring)
    rad = rr[ringsk == 1]
    assert rad.min() >= 16.5 and rad.max() <= 19.5               # the medial circle of the annulus (radius 18) within 1.5 px


def test_synth_las_point_records_match_the_las_layout(tmp_path):
    """synth.las_point_records (what `bench.py --points las` keeps in pinned host memory) lays a point out like an ASPRS LAS 1.2 format-0
    record: the coordinate and intensity fields equal the oracle writer's for the same points, record length 20, and the numpy LAS reader
    decodes a file made of these records to the quantised coordinates."""
    from lanemapping_amd import synth
    from oracle import las_ref
    pts = synth.las_points(9, 5000)
    rec = synth.las_point_records(pts, 1e-3).reshape(-1, 20)
    path = str(tmp_path / 'a.las')
    q = las_ref.write_las(path, pts[:, :3].astype(np.float64), pts[:, 3], point_format=0, scale=(1e-3,) * 3, offset=(0., 0., 0.))
    data = np.fromfile(path, np.uint8)
    off = int(np.frombuffer(data[96:100].tobytes(), '<u4')[0])
    ref = data[off:off + 20 * len(pts)].reshape(-1, 20)
    assert las_ref.RECORD_LEN[0] == 20 and np.array_equal(rec[:, :14], ref[:, :14])        # X, Y, Z (int32), intensity (u16)
    assert np.array_equal(rec[:, 0:12].copy().view('<i4').reshape(-1, 3), q)
    # a file whose records are the synth ones decodes to the same points through the reader
    with open(path, 'r+b') as f:
        f.seek(off)
        f.write(rec.tobytes())
    got = las_ref.read_las_ref(path, normalise=False)
    assert np.array_equal(got[:, :3], q * 1e-3) and np.array_equal(got[:, 3], np.floor(pts[:, 3]).astype(np.float64))


def test_bench_second_line_never_breaks_the_headline(monkeypatch):
    """bench.second_line: whatever happens to the child run (non-zero exit, no JSON line, an exception), the headline gets an `error` entry
    instead of an exception; a good child line is reduced to the declared fields."""
    import json
    import subprocess
    import types
    import bench
    args = types.SimpleNamespace(steps=100, no_graphs=False, streams=None)
    seen = {}

    def fake(rc, out, err=''):
        def run(cmd, **kw):
            seen['cmd'], seen['env'] = cmd, kw['env']
            return types.SimpleNamespace(returncode=rc, stdout=out, stderr=err)
        return run
    monkeypatch.setattr(subprocess, 'run', fake(1, '', 'boom'))
    assert 'exit code 1' in bench.second_line(args)['error'] and 'boom' in bench.second_line(args)['error']
    assert seen['env']['LANEMAP_WINO_SPLIT'] == '1' and '--no-second-line' in seen['cmd'] and seen['cmd'][seen['cmd'].index('--steps') + 1] == '40'
    monkeypatch.setattr(subprocess, 'run', fake(0, 'no json here'))
    assert 'error' in bench.second_line(args)

    def boom(cmd, **kw):
        raise subprocess.TimeoutExpired(cmd, 900)
    monkeypatch.setattr(subprocess, 'run', boom)
    assert 'TimeoutExpired' in bench.second_line(args)['error']
    line = {'value': 460.0, 'unit': 'tiles/s', 'ms_per_step': 34.7, 'steps': 40, 'dtype': 'f16x2-split', 'config': {'windows_tiles_per_s': None, 'stream_check': 'ok', 'raster_check': 'ok'},
            'roofline': {'per_kernel': {}, 'kernel_ms_per_step': 30.0, 'winograd_ms_per_step': 26.0, 'frac': 0.2, 'scope': 's'}}
    monkeypatch.setattr(subprocess, 'run', fake(0, 'noise\n' + json.dumps(line) + '\n'))
    sl = bench.second_line(args)
    assert sl['value'] == 460.0 and sl['steps'] == 40 and 'NOT bit-identical' in sl['what'] and sl['roofline']['frac'] == 0.2 and 'error' not in sl
