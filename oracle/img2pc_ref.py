"""Oracle (TEST INFRASTRUCTURE ONLY) for SURVEY §8f row f1: BEV polylines -> LAS frame.

numpy restatement of baseline/utils/coor_img2pc.py: multiplyQuanternion :22-32, rotateByQuanternion3D :40-53,
LeastSuqare :59-73, modify_empty_pixel_elevation (roi form) :94-122, transform_coordinate_from_img_2_pc :127-183.
Pinned: tests/golden/g12_img2pc.npz is produced by importing the reference module (tests/golden/make_golden.py g12) and
this restatement matches it bit for bit (tests/test_oracle_golden.py).
"""
import numpy as np

EPS = 1e-6


def _qmul(a, b):
    o = np.zeros(4)
    o[0] = a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3]
    o[1] = a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2]
    o[2] = a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1]
    o[3] = a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]
    return o


def rotate(quan, v):
    """q v q* / |q|  (:40-53)."""
    quan = np.asarray(quan, dtype=np.float64)
    norm = np.sqrt(np.sum(np.square(quan)))
    inv = quan.copy()
    inv[1:4] *= -1.
    inv /= norm
    return _qmul(_qmul(quan, np.array([0., v[0], v[1], v[2]])), inv)[1:]


def least_square(X, Y):
    N = len(Y)
    p = N * sum(X * Y) - sum(X) * sum(Y)
    q = N * sum(X * X) - sum(X) * sum(X)
    w = 0. if abs(q) < EPS else p / q
    return w, sum(Y - w * X) / N


def fill_vertex_elevation(img, seqs, lens):
    """roi form of modify_empty_pixel_elevation (:94-122); edits img in place."""
    H, W, _ = img.shape
    for l in range(seqs.shape[0]):
        for k in range(lens[l]):
            h, w = int(seqs[l, k, 0]), int(seqs[l, k, 1])
            if (h == 0 and w == 0) or np.sum(img[h, w, :]) > 1:
                continue
            step, total = 1, 0
            while total < 1:
                win = img[max(h - step, 0):min(h + step, H), max(w - step, 0):min(w + step, W), :]
                total = np.sum(win)
                if total > 0:
                    valid = len(np.where(np.sum(win, axis=2) > 0)[0])
                    img[h, w, 1] = np.sum(win[:, :, 1]) / valid
                else:
                    step += 1
    return img


def img_to_pc_ref(params, img_seqs, lens, bev_img):
    img_seqs = np.asarray(img_seqs, dtype=np.float64)
    L, V, _ = img_seqs.shape
    out = np.zeros((L, V, 3))
    out[:, :, 0] = img_seqs[:, :, 0] * params['img_reso'][0] + params['bev_img_offset'][0]
    out[:, :, 1] = img_seqs[:, :, 1] * params['img_reso'][1] + params['bev_img_offset'][1]
    img = fill_vertex_elevation(np.array(bev_img), img_seqs, lens)
    out[:, :, 2] = img[img_seqs[:, :, 0].astype(int), img_seqs[:, :, 1].astype(int), 1] * params['ele_reso'] + params['local_min_ele']
    for l in range(L):
        idx = np.arange(lens[l])
        w, b = least_square(idx, np.array(out[l, :lens[l], 2]))
        out[l, :lens[l], 2] = w * idx + b
    q = np.array(params['las_rotation_trans_quan'][3:])
    t = np.array(params['las_rotation_trans_quan'][0:3])
    for l in range(L):
        for k in range(V):
            out[l, k, :] = rotate(q, out[l, k, :])
            out[l, k, :] += t
    return out + np.array(params['las_read_offset'])
