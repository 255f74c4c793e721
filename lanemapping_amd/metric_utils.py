"""Evaluation metrics of the reference's test loop, SURVEY §8f row f3 (the part that needs no skeletonisation).

  cal_coor_measures(arr_label, arr_pred, 'conf', offset_thre)   <- baseline/utils/metric_utils.py:47-64 + :112-166
      vertex precision / recall / F1: a predicted vertex is a true positive when some GT lane has a vertex on the same row
      strictly inside (col - r, col + r) clipped to [0, 1151]; symmetric for recall.
  eval_metric_endp_detector(endp_pred, endp_gt, r_thre)         <- :483-513
      endpoint precision / recall / F1 with nearest-neighbour distance < r_thre (the reference uses a cKDTree; the point
      sets hold a few dozen pixels, so an exact all-pairs distance gives the same counts).
  eval_metric_line_segmentor(seg_result, mask, bi_seg, semantics, buff)   <- :415-481
      semantic-line precision / recall / F1 on skeletons: the predicted class map is thinned (Lee-Kashyap-Chu, the algorithm behind
      skimage's skeletonize(method='lee'); lm_skeletonize_lee_2d, host C++) and skeleton / ground-truth pixels are matched by
      nearest-neighbour distance < buff (scipy cKDTree like the reference).  SEMANTIC-LINE F1 IS PARITY UNPINNED: skimage and cv2 are not
      in the build image, so neither the thinning nor the cv2.line raster of the predicted map (hostpost.raster_semantic_map) can be
      compared with the libraries the reference calls; both restate the published algorithms and are held by known-answer tests
      (tests/test_metrics_io_cpu.py: published thinning shapes, the OpenCV LineIterator table) and a literal 3-D restatement.
Same return tuples as the reference.  Host-side: these run once per tile, far from the hot path.
"""
import ctypes as C

import numpy as np

EPS = 1e-16      # baseline/utils/metric_utils.py:11
_W = 1152


def _hits(a, b, r):
    """For every positive entry of a [La, H]: does any row-mate in b [Lb, H] fall strictly inside its clipped buffer?"""
    lo = np.maximum(0, a - r)[:, None, :]
    hi = np.minimum(_W - 1, a + r)[:, None, :]
    inside = (b[None, :, :] > lo) & (b[None, :, :] < hi)
    return inside.any(axis=1) & (a > 0)


def calc_coor_measures_conf_metric2(arr_label, arr_pred, buff_radius=2):
    arr_label, arr_pred = np.asarray(arr_label), np.asarray(arr_pred)
    n_pred_pts = int((arr_pred > 0).sum())
    n_gt_pts = int((arr_label > 0).sum())
    tp = int(_hits(arr_pred, arr_label, buff_radius).sum())
    dg = int(_hits(arr_label, arr_pred, buff_radius).sum())
    return tp, n_pred_pts, dg, n_gt_pts


def calc_coor_measures_cls(arr_label, arr_pred, offset_thre=8):
    dist = np.abs(arr_label - arr_pred)
    tp = int(((dist < offset_thre) & (arr_label > 0) & (arr_pred > 0)).sum())
    fp = int((((dist >= offset_thre) | (arr_label < 0)) & (arr_pred > 0)).sum())
    fn = int((((dist >= offset_thre) | (arr_pred < 0)) & (arr_label > 0)).sum())
    return tp, fp, fn


def cal_coor_measures(arr_label, arr_pred, mode='conf', offset_thre=8):
    if mode != 'conf':
        raise NotImplementedError("cal_coor_measures: only mode='conf' is well defined (the reference's other branch "
                                  'unpacks three values into four names)')
    TP, num_seg_pts, DG, num_gt_pts = calc_coor_measures_conf_metric2(arr_label, arr_pred, buff_radius=offset_thre)
    acc = TP / (num_seg_pts + EPS)
    recall = DG / (num_gt_pts + EPS)
    f1 = 2.0 * acc * recall / (acc + recall + EPS)
    return acc, recall, f1, TP, num_seg_pts, DG, num_gt_pts


def _nearest(a, b):
    d = np.sqrt(((a[:, None, :].astype(np.float64) - b[None, :, :]) ** 2).sum(-1))
    return d.min(axis=1)


def eval_metric_endp_detector(endp_pred, endp_gt, r_thre=10):
    gt = np.argwhere(np.asarray(endp_gt) > 0.99)
    pr = np.argwhere(np.asarray(endp_pred) > 0.99)
    TPs = DGs = seg_pts = gt_pts = 0
    if len(gt) > 0 and len(pr) > 0:
        DGs = int((_nearest(gt, pr) < r_thre).sum())
        gt_pts = len(gt)
        TPs = int((_nearest(pr, gt) < r_thre).sum())
        seg_pts = len(pr)
    acc = TPs / seg_pts if seg_pts > 0 else 0.
    rec = DGs / gt_pts if gt_pts > 0 else 0.
    f = 2 * rec * acc / (acc + rec) if (acc + rec) > 0. else 0
    return acc, rec, f, TPs, seg_pts, DGs, gt_pts


def skeletonize_lee(image):
    """2-D binary image (nonzero = object) -> uint8 skeleton (0 / 1), Lee-Kashyap-Chu thinning (csrc/skeleton.cpp)."""
    from ._lib import lib
    img = np.ascontiguousarray((np.asarray(image) != 0).astype(np.uint8))
    assert img.ndim == 2
    if lib().lm_skeletonize_lee_2d(img.ctypes.data_as(C.c_void_p), img.shape[0], img.shape[1]) < 0:
        raise ValueError('skeletonize_lee: bad image')
    return img


def _match_counts(graph_pts, gt_pts, thre):
    """(TPs, seg_pts, DGs, gt_pts) of one class: skeleton pixels within `thre` of a GT pixel, GT pixels within `thre` of the skeleton."""
    import scipy.spatial
    gt_tree = scipy.spatial.cKDTree(gt_pts)
    graph_tree = scipy.spatial.cKDTree(graph_pts)
    graph_dds, _ = graph_tree.query(gt_pts, k=1)
    gt_acc_dds, _ = gt_tree.query(graph_pts, k=1)
    return int((gt_acc_dds < thre).sum()), len(gt_acc_dds), int((graph_dds < thre).sum()), len(graph_dds)


def eval_metric_line_segmentor(seg_result, mask, bi_seg=True, semantics=2, buff=10):
    """Reference semantics (:415-481), including its accounting when one side is empty: no GT pixels -> every skeleton pixel is a
    false positive; GT but no skeleton -> every GT pixel is missed; bi_seg matches all non-zero pixels at once, otherwise the
    classes 1..semantics are thinned and matched one by one and the counts add up."""
    seg_result, mask = np.asarray(seg_result), np.asarray(mask)
    TPs = DGs = seg_pts = gt_pts = 0
    classes = [None] if bi_seg else list(range(1, semantics + 1))
    for cls in classes:
        skel = skeletonize_lee(seg_result.astype(np.int8) if cls is None else (seg_result == cls))
        gt = np.argwhere(mask != 0) if cls is None else np.argwhere(mask == cls)
        graph = np.argwhere(skel != 0)
        if len(gt) > 0:
            if len(graph) > 0:
                tp, sp, dg, gp = _match_counts(graph, gt, buff)
                TPs, seg_pts, DGs, gt_pts = TPs + tp, seg_pts + sp, DGs + dg, gt_pts + gp
            else:
                gt_pts += len(gt)
        else:
            seg_pts += len(graph)
    acc = TPs / seg_pts if seg_pts > 0 else 0.
    rec = DGs / gt_pts if gt_pts > 0 else 0.
    f = 2 * rec * acc / (acc + rec) if acc * rec else 0
    return acc, rec, f, TPs, seg_pts, DGs, gt_pts
