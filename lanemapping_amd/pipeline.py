"""End-to-end tile pipeline (the bench / runner hot loop): HBM-resident tiles -> polylines.

GPU: Detector1stage.forward_raw + decode kernels (one stream).  Host: endpoint clustering + polyline assembly
in C++ (ctypes releases the GIL), fanned out over a thread pool, one task per tile, and overlapped with the
GPU work of the next batch.  Replaces the per-batch body of Runner.infer_lane_coordinate_endpoint_semantics
(reference engine/runner.py:725-828) minus metrics / overlays.
"""
import collections
import os
import threading
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

from . import decode, hostpost, ops, torch_ops, trace
from ._lib import LanemapHipError


class TilePipeline:
    def __init__(self, net, host_threads=8, use_graph=None, with_decode_endp=False):
        """use_graph (default: env LANEMAP_GRAPHS=1): capture the device part of a batch (network + decode kernels, ~350 launches) into
        one HIP graph per input shape and replay it - the host then spends one launch per batch instead of ~9 ms of enqueue work.
        Same kernels, same arguments, same stream order: bit-identical outputs (test_tile_pipeline_graph_replay_bit_identical)."""
        self.with_decode_endp = bool(with_decode_endp)   # tile results become (lanes, kept endpoints, the decode's endpoints before the filter)
        self.use_graph = (os.environ.get('LANEMAP_GRAPHS', '0') != '0') if use_graph is None else bool(use_graph)
        self._graphs = collections.OrderedDict()      # (shape, dtype, device) -> (graph, static input, outputs, weight key); LRU, MAX_GRAPHS entries
        self.net = net
        self.cfg = net.cfg
        self.rowref = net.cfg.heads.type == 'RowSharNotReducRef'
        if net.cfg.heads.type not in ('ColumnProposal2', 'RowSharNotReducRef'):
            raise NotImplementedError(f'TilePipeline has no decode / polyline tail for heads.type={net.cfg.heads.type!r}')
        self.pool = ThreadPoolExecutor(max_workers=host_threads)
        self._pending = None
        self._slots, self._slot_next = [], 0           # ring of pinned staging blocks
        self.host_seconds = 0.0      # accumulated wall time of the per-tile host tasks (all threads) and their count
        self.host_tiles = 0
        self._stats_lock = threading.Lock()

    HOST_SLOTS = 3          # pinned staging blocks per pipeline (batch k is being filled while the pool still reads batch k - 1)

    def _host_slot(self, nbytes):
        """Next pinned staging block of the ring (allocated once per pipeline, grown when a larger batch arrives); waits for the host
        tasks that still read it - with a pipeline depth of 1 they finished two batches ago."""
        if not self._slots:
            self._slots = [{'block': None, 'futs': []} for _ in range(self.HOST_SLOTS)]
        slot = self._slots[self._slot_next]
        self._slot_next = (self._slot_next + 1) % self.HOST_SLOTS
        # the ring is safe for a pipeline depth of 1 (HOST_SLOTS >= depth + 2): the batch whose host tasks have not been created yet
        # (self._pending) must never own the block that is about to be overwritten
        assert self._pending is None or self._pending[4] is not slot, 'pinned staging ring too short for the pipeline depth'
        if slot['futs']:
            from concurrent.futures import wait
            wait(slot['futs'])                            # (the caller sees errors / cancellations through its own futures)
        slot['futs'] = []
        if slot['block'] is None or slot['block'].numel() < nbytes:
            slot['block'] = torch.empty(max(nbytes, 1), dtype=torch.uint8, pin_memory=True)
        return slot

    def _gpu_stage(self, proj):
        if self.use_graph and torch.is_tensor(proj):      # (a tile tensor: the LiDAR path sizes its launches on the host and cannot be captured)
            (block, layout), keep, crop = self._replay(proj)
        else:
            (block, layout), keep, crop = self._device_part(proj)
        # device -> pinned host: ONE copy of the packed block; the staging memory comes from the ring (no pinned allocation per batch)
        total = block.numel()
        slot = self._host_slot(total)
        hb = slot['block']
        hb[:total].copy_(block, non_blocking=True)
        host = {k: hb[o:o + nb].view(dt).view(shape) for k, o, nb, dt, shape in layout}
        ev = torch.cuda.Event()
        ev.record()
        return host, ev, (keep, block), crop, slot

    def _pack(self, dev):
        """The tensors the host tasks read, gathered into one device block (ops.pack_readback: one kernel, capturable)."""
        names = list(dev)
        if len(names) == 1 and dev[names[0]].is_contiguous():          # (config 4: one tensor - nothing to gather)
            t = dev[names[0]]
            return t.view(-1).view(torch.uint8), [(names[0], 0, t.numel() * t.element_size(), t.dtype, tuple(t.shape))]
        ts = [dev[k].contiguous() for k in names]
        block, segs = ops.pack_readback(ts)
        end = max(o + nb for o, nb in segs)
        return block[:end], [(k, o, nb, t.dtype, tuple(t.shape)) for k, t, (o, nb) in zip(names, ts, segs)]

    MAX_GRAPHS = 4          # captured graphs kept per pipeline (each one pins a private activation pool: ~1 GB per tile of its batch)

    def _weights_key(self):
        """What a captured graph is valid for: the parameter identities / versions of every module that packs weights.  A graph bakes
        the packed-weight and workspace pointers in and never re-runs PackedModule.packed(), so load_ckpt / load_state_dict / .to() /
        an in-place edit must force a recapture (the eager path repacks through the same key)."""
        from .packing import PackedModule
        mods = self.__dict__.get('_packed_mods')
        if mods is None:
            mods = self.__dict__['_packed_mods'] = [m for m in self.net.modules() if isinstance(m, PackedModule)]
        return tuple(m.param_key() for m in mods)

    def clear_graphs(self):
        """Drop every captured graph (and the activation pools they pin)."""
        self._graphs.clear()

    def _replay(self, proj):
        """HIP-graph path: static input / output buffers per (shape, dtype); the first batch of a shape runs once eagerly (lazy
        initialisation: weight packing, kernel attributes, workspaces) and is then captured on a stream of its own."""
        key = (tuple(proj.shape), proj.dtype, proj.device)
        wkey = self._weights_key()
        ent = self._graphs.get(key)
        if ent is not None and ent[3] != wkey:        # the weights changed under the graph: recapture
            del self._graphs[key]
            ent = None
        if ent is None:
            static_in = torch.empty_like(proj)
            static_in.copy_(proj)
            self._device_part(static_in)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=torch.cuda.Stream(device=proj.device)):
                out = self._device_part(static_in)
            ent = (graph, static_in, out, wkey)
            self._graphs[key] = ent
            while len(self._graphs) > self.MAX_GRAPHS:
                self._graphs.popitem(last=False)
        self._graphs.move_to_end(key)
        graph, static_in, out, _ = ent
        static_in.copy_(proj, non_blocking=True)
        graph.replay()
        return out

    def _device_part(self, proj):
        """Every device-side launch of a batch (no host reads, no pinned allocations: capturable): raw net outputs + decode kernels + the
        gather of what the host reads into one block.  Returns ((block, layout), keep-alive, crop width)."""
        heads, cfg = self.net.heads, self.cfg
        # a [B,3,H,W] tile tensor (FPN path), a list of [N_i,4] point tensors (sparse-conv LiDAR path, config 5) or a batch dict
        batch = proj if isinstance(proj, dict) else ({'points': list(proj)} if isinstance(proj, (list, tuple)) else {'proj': proj})
        raw = self.net.forward_raw(batch)
        if self.rowref:
            # config 4 (reference row_shared_not_reduc_ref.py:334-393, :487-516): argmax columns on the device, per-lane tracing on the host
            dev = {'col': heads.decode_columns(raw)}
            keep = (raw, dev)
            crop = 0
        else:
            # decode kernels as dispatcher-visible custom ops (torch.ops.lanemap_hip.*, torch_ops.py)
            trace.push('decode')
            prop_conf, v_ext, cls_conf, cls_idx, cls_offset = torch_ops.decode_proposals(
                raw['proposal_conf'], raw['ext2'], raw['cls2'], raw['offset2'], float(cfg.exist_thre), heads.prop_width, heads.prop_half_buff)
            orient = torch_ops.decode_orient(raw['orient'])
            sem, biseg, rows = torch_ops.decode_semantic(raw['semantic_seg'], float(cfg.coor_thre))
            idx, score, status = torch_ops.endp_topk(raw['endp_est'], decode.TOPK, decode.CLIP)
            trace.pop()
            dev = {'prop_conf': prop_conf, 'v_ext': v_ext, 'cls_offset': cls_offset, 'rows': rows, 'idx': idx, 'status': status}
            keep = (raw, sem, biseg, orient, cls_conf, cls_idx, dev)      # keep device buffers alive until the copies land
            crop = raw['endp_est'].shape[-1]
        return self._pack(dev), keep, crop

    def _tile_task(self, host, b, crop_w):
        t0 = time.perf_counter()
        if self.rowref:
            heads = self.net.heads
            cols = heads.lines_from_columns(host['col'][b].numpy(), heads.row_size)       # [L,144] column px (<= 0: none)
            lanes = np.full((72, 144, 2), -1.0)                 # the [72,144,2] block the JSON writer / all-gather use
            lanes[:, :, 1] = 0.0
            lanes[:cols.shape[0], :, 0] = cols
            lanes[:cols.shape[0], :, 1] = (cols > 0).astype(np.float64)
            kept = pts = np.zeros((0, 2), dtype=np.int32)
        else:
            pts, _ = hostpost.cluster_endpoints(host['idx'][b].numpy(), crop_w=crop_w, clip=decode.CLIP,
                                                k0=self.net.heads.num_cls * 2 * 10, k_max=500)
            lanes, kept = hostpost.assemble_polylines(host['prop_conf'][b].numpy(), host['v_ext'][b].numpy(),
                                                      host['cls_offset'][b].numpy(), host['rows'][b].numpy(), pts,
                                                      self.cfg.proposal_obj_thre)
        dt = time.perf_counter() - t0
        with self._stats_lock:                              # pool threads finish tiles concurrently
            self.host_seconds += dt
            self.host_tiles += 1
        return (lanes, kept, pts) if self.with_decode_endp else (lanes, kept)

    def _finish(self, pending):
        host, ev, keep, W, slot = pending
        ev.synchronize()
        if not self.rowref and int(host['status'].max()) != 0:
            raise LanemapHipError('endpoint top-K candidate overflow')   # guard only: the compaction is tie-safe (<= K candidates)
        B = next(iter(host.values())).shape[0]
        futs = [self.pool.submit(self._tile_task, host, b, W - 2 * decode.CLIP) for b in range(B)]
        slot['futs'] = futs                               # the staging block is reused only after these have run
        return futs

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
