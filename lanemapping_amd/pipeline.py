"""End-to-end tile pipeline (the bench / runner hot loop): HBM-resident tiles -> polylines.

GPU: Detector1stage.forward_raw + decode kernels (one stream).  Host: endpoint clustering + polyline assembly
in C++ (ctypes releases the GIL), fanned out over a thread pool, one task per tile, and overlapped with the
GPU work of the next batch.  Replaces the per-batch body of Runner.infer_lane_coordinate_endpoint_semantics
(reference engine/runner.py:725-828) minus metrics / overlays.
"""
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import decode, hostpost, ops
from ._lib import LanemapHipError


class TilePipeline:
    def __init__(self, net, host_threads=8):
        self.net = net
        self.cfg = net.cfg
        if net.cfg.heads.type != 'ColumnProposal2':
            raise NotImplementedError(f"TilePipeline drives the ColumnProposal2 decode / polyline tail; for heads.type="
                                      f"{net.cfg.heads.type!r} call the net directly (Detector1stage.forward returns its lane_maps)")
        self.pool = ThreadPoolExecutor(max_workers=host_threads)
        self._pending = None
        self.host_seconds = 0.0      # accumulated wall time of the per-tile host tasks (all threads) and their count
        self.host_tiles = 0

    def _gpu_stage(self, proj):
        heads, cfg = self.net.heads, self.cfg
        # a [B,3,H,W] tile tensor (FPN path), a list of [N_i,4] point tensors (sparse-conv LiDAR path, config 5) or a batch dict
        batch = proj if isinstance(proj, dict) else ({'points': list(proj)} if isinstance(proj, (list, tuple)) else {'proj': proj})
        raw = self.net.forward_raw(batch)
        prop_conf, v_ext, cls_conf, cls_idx, cls_offset = ops.decode_proposals(
            raw['proposal_conf'], raw['ext2'], raw['cls2'], raw['offset2'], cfg.exist_thre, heads.prop_width, heads.prop_half_buff)
        orient = ops.decode_orient(raw['orient'])
        sem, biseg, rows = ops.decode_semantic(raw['semantic_seg'], cfg.coor_thre)
        idx, score, status = ops.endp_topk(raw['endp_est'], K=decode.TOPK, clip=decode.CLIP)
        dev = {'prop_conf': prop_conf, 'v_ext': v_ext, 'cls_offset': cls_offset, 'rows': rows, 'idx': idx, 'status': status}
        host = {k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True) for k, v in dev.items()}
        for k in dev:
            host[k].copy_(dev[k], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        keep = (raw, sem, biseg, orient, cls_conf, cls_idx, dev)      # keep device buffers alive until the copies land
        return host, ev, keep, raw['endp_est'].shape[-1]

    def _tile_task(self, host, b, crop_w):
        t0 = time.perf_counter()
        pts, _ = hostpost.cluster_endpoints(host['idx'][b].numpy(), crop_w=crop_w, clip=decode.CLIP,
                                            k0=self.net.heads.num_cls * 2 * 10, k_max=500)
        lanes, kept = hostpost.assemble_polylines(host['prop_conf'][b].numpy(), host['v_ext'][b].numpy(),
                                                  host['cls_offset'][b].numpy(), host['rows'][b].numpy(), pts,
                                                  self.cfg.proposal_obj_thre)
        self.host_seconds += time.perf_counter() - t0      # (benign race between pool threads: statistics only)
        self.host_tiles += 1
        return lanes, kept

    def _finish(self, pending):
        host, ev, keep, W = pending
        ev.synchronize()
        if int(host['status'].max()) != 0:
            raise LanemapHipError('endpoint top-K candidate overflow (too many tied scores)')
        B = host['idx'].shape[0]
        return [self.pool.submit(self._tile_task, host, b, W - 2 * decode.CLIP) for b in range(B)]

    def submit(self, proj):
        """Enqueue one batch; returns the futures of the PREVIOUS batch's tiles (software pipeline depth 1)."""
        with torch.no_grad():
            new = self._gpu_stage(proj)
        futs = self._finish(self._pending) if self._pending is not None else []
        self._pending = new
        return futs

    def flush(self):
        futs = self._finish(self._pending) if self._pending is not None else []
        self._pending = None
        return futs

    def run_batch(self, proj):
        """Synchronous convenience: one batch in, list of (lanes [72,144,2] f64, endpoints [k,2]) out."""
        assert self._pending is None
        self.submit(proj)
        return [f.result() for f in self.flush()]
