"""Inference runner: the `test_gpu_0.py` / `Runner` entry of the reference (hot-path subset + the evaluation loop).

  load_config_and_runner(path, gpus)                 <- baseline/engine/runner.py:57-66 (log_dir + '/vis', work_dirs = log_dir/<dataset.train.type>)
  Runner.load_ckpt(path)                             <- :399-401 (strict, 'module.'-prefixed keys accepted)
  Runner.infer_lane_coordinate_endpoint_semantics()  <- :690-867, the reference's signature: the split `mode_data` (default cfg.dataset.test)
                                                        is listed the reference's way (datasets.py: <data_root>/<data_split_file>[mode] ->
                                                        <data_root>/cropped_tiff/<stem>.png), every tile -> <work_dirs>/<image_name[0:11]>.json
                                                        via save_lane_seq_2d when write_lane_vertex, and with gt_avail the loop's
                                                        coordinate / endpoint / semantic counters and its nine P / R / F1 lines
  Runner.infer_lane_coordinate()                     <- :606-687 (the K-Lane / RowRef entry, config 4: coordinate measures only)
  Runner.infer_lane_geometry_segmentation_segmentor()<- :945-1036 (Segmentor config)
  Runner.infer_las_to_map()                          <- the offline chain LAS -> BEV -> polylines -> LAS frame -> merged map
                                                        (read_las, Las2BEV, Runner, coor_img2pc.py, merge_lines.py) in one call
Deviations, all deliberate: (1) tiles are walked SORTED by stem, not in the seeded shuffle of the reference's test list (SURVEY C13;
per-tile results and the summed counters do not depend on the order; datasets.load_datadir(shuffle_seed=cfg.seed) gives the reference's
order); (2) `mode_view=True` is accepted and ignored with one notice: the cv2 overlays (:793-822) are not results (OUT);
(3) keyword-only extras `tiles=` (explicit list / directory of PNG tiles instead of a split: no labels, so no evaluation),
`batch_size=`, `work_dirs=`.  Unknown keywords raise TypeError, an empty tile list raises ValueError.
Tiles are PNG files (load_img contract, datasets/laserlane_proposals.py:85-98): decoded on the host by the library's own PNG reader and DEFLATE decoder (png_io / csrc/inflate.h, on the host thread pool),
converted u8 -> f32/255 on the GPU (lm_tile_ingest_u8).  With torch.distributed initialised, tiles are sharded
over the ranks (lanemapping_amd/shard.py) and rank 0 writes every file after one all-gather per batch.
"""
import glob
import os
import random

import numpy as np
import torch

from . import io_utils, ops, shard
from .boundary import load_config, load_reference_checkpoint
from .pipeline import TilePipeline
from .registry import build_net


def load_config_and_runner(path_config, gpus='0'):
    """baseline/engine/runner.py:57-66.  `gpus` is the reference's GPUS_EN string (test_gpu_0.py:7-9): one id -> a `Runner` on this
    process's GPU (cuda:$LOCAL_RANK, default 0; under torchrun every rank calls this with one id); several ids -> a `MultiGpuRunner`
    (runner_ranks.py) that fans every inference call out over one fresh process per listed GPU - the place of the reference's
    DataParallel(device_ids=range(cfg.gpus)) (:103-104).  A malformed list raises ValueError, more ids than visible GPUs RuntimeError."""
    from .runner_ranks import MultiGpuRunner, parse_gpus
    ids = parse_gpus(gpus)
    cfg = load_config(path_config)
    cfg.log_dir = cfg.log_dir + '/vis'
    os.makedirs(cfg.log_dir, exist_ok=True)
    cfg.work_dirs = cfg.log_dir + '/' + cfg.dataset.train.type
    os.makedirs(cfg.work_dirs, exist_ok=True)
    cfg.gpus = len(ids)
    if len(ids) > 1:
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            raise RuntimeError(f'gpus={gpus!r} inside an initialised torch.distributed job: every rank of a launcher-started job drives ONE '
                               f'GPU - pass one id per rank (the rank\'s device is cuda:$LOCAL_RANK)')
        return cfg, MultiGpuRunner(cfg, ids)
    return cfg, Runner(cfg)


EPS = 1e-16      # baseline/engine/runner.py:30


def _prf(tp, dets, dg, gts):
    """precision / recall / F1 from the loop's counters (runner.py:843-857)."""
    pre = tp / (dets + EPS)
    rec = dg / (gts + EPS)
    f1 = 2. * pre * rec / (pre + rec) if (pre + rec) > 0. else 0.
    return pre, rec, f1


class Runner:
    def __init__(self, cfg, device=None):
        self.cfg = cfg
        seed = int(cfg.get('seed', 2021))
        torch.manual_seed(seed)
        np.random.seed(seed)
        random.seed(seed)
        self.device = torch.device(device or ('cuda:%d' % int(os.environ.get('LOCAL_RANK', 0))))
        if self.device.type == 'cuda':
            # the C library launches on the CURRENT HIP device / its current stream (ops._stream): one process drives one GPU
            torch.cuda.set_device(self.device)
        self.net = build_net(cfg).eval().to(self.device)

    def load_ckpt(self, path_ckpt):
        return load_reference_checkpoint(self.net, path_ckpt, strict=True)

    # ------------------------------------------------------------------------------------------------ input
    @staticmethod
    def list_tiles(source):
        if isinstance(source, (list, tuple)):
            return sorted(source)
        return sorted(glob.glob(os.path.join(source, '*.png')))

    def _decode_batch(self, paths, slot):
        """Host part of the tile ingest: the library's PNG reader inflates the batch on host threads into one of two pinned
        [n,H,W,C] buffers (pure C, no GIL, so it can run one batch ahead on a helper thread)."""
        from .png_io import read_png_batch, png_info
        with open(paths[0], 'rb') as f:
            h, w, c = png_info(f.read(64))
        pinned = self.__dict__.setdefault('_pinned', {})      # slot -> (pinned uint8 buffer, event of the last copy out of it)
        buf, ev = pinned.get(slot, (None, None))
        if ev is not None:
            ev.synchronize()                                   # the previous copy out of this buffer has finished
        if buf is None or tuple(buf.shape[1:]) != (h, w, c) or buf.shape[0] < len(paths):
            buf = torch.empty((len(paths), h, w, c), dtype=torch.uint8, pin_memory=self.device.type == 'cuda')
        view = buf[:len(paths)]
        read_png_batch(paths, threads=int(self.cfg.get('host_threads', 8)), out=view.numpy())
        pinned[slot] = (buf, None)
        return view

    def _to_device(self, view, slot):
        u8 = view.to(self.device, non_blocking=True)
        if self.device.type == 'cuda':
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.device))
            self._pinned[slot] = (self._pinned[slot][0], ev)
        if u8.shape[-1] < 3:                                   # greyscale tiles: replicate into the 3 input channels
            u8 = u8[..., :1].expand(-1, -1, -1, 3).contiguous()
        elif u8.shape[-1] > 3:                                 # RGBA: the reference keeps the first three channels (load_img)
            u8 = u8[..., :3].contiguous()
        return u8           # the stem kernel takes the u8 HWC tile and applies u8 / 255 itself (ops.stem; == ops.tile_ingest + f32 stem)

    def _load_batch(self, paths):
        return self._to_device(self._decode_batch(paths, 0), 0)

    def _batches(self, paths, B):
        """Yields the device tensor of every batch; batch i+1 is decoded on a helper thread while batch i is enqueued and runs."""
        from concurrent.futures import ThreadPoolExecutor
        chunks = [paths[i:i + B] for i in range(0, len(paths), B)]
        if not chunks:
            return
        with ThreadPoolExecutor(max_workers=1) as pool:
            fut = pool.submit(self._decode_batch, chunks[0], 0)
            for k in range(len(chunks)):
                view = fut.result()
                if k + 1 < len(chunks):
                    fut = pool.submit(self._decode_batch, chunks[k + 1], (k + 1) & 1)
                yield self._to_device(view, k & 1)

    # ------------------------------------------------------------------------------------------------ inference
    def _entries(self, mode_data, tiles):
        """The tiles of a run, sorted by stem: [(image_name, png path, dataset entry | None)]."""
        from . import datasets
        if tiles is not None:
            paths = self.list_tiles(tiles)
            ents = [(os.path.splitext(os.path.basename(p))[0][0:11], p, None) for p in paths]
        else:
            split = self.cfg.dataset.test if mode_data is None else mode_data
            ents = sorted(datasets.split_entries(split, self.cfg), key=lambda e: e['stem'])
            # the reference's stems keep the dot of '<stem>.json' (laserlane_proposals.py:534) and are cut to 11 characters (:76)
            ents = [((e['stem'] + '.')[0:11], e['image'], e) for e in ents]
        if not ents:
            raise ValueError('Runner: no tiles to process (' + (f'tiles={tiles!r}' if tiles is not None else 'the split lists none') + ')')
        missing = [p for _, p, _ in ents if not os.path.isfile(p)]
        if missing:
            raise FileNotFoundError(f'Runner: {len(missing)} of {len(ents)} tiles do not exist, first: {missing[0]}')
        return ents

    def _view_notice(self, mode_view):
        if mode_view and not self.__dict__.get('_view_noticed'):
            self._view_noticed = True
            print('lanemapping_amd.Runner: mode_view=True - the *_source / *_offset / *_seg / *_gt PNG overlays of the reference are '
                  'not produced (cv2 drawing is outside the hot path); results, JSON files and metrics are unaffected')

    def _infer(self, path_ckpt, mode_data, mode_view, gt_avail, write_lane_vertex, measures, tiles, batch_size, work_dirs):
        """Shared body of the two detector entries.  measures: subset of ('coor', 'endp', 'semantic') | ('klane',)."""
        from . import datasets, hostpost, metric_utils
        if path_ckpt:
            self.load_ckpt(path_ckpt)
        self._view_notice(mode_view)
        ents = self._entries(mode_data, tiles)
        if tiles is not None:
            gt_avail = False                                     # an explicit tile list carries no labels
        out_dir = work_dirs or self.cfg.get('work_dirs', './work_dirs')
        if write_lane_vertex:
            os.makedirs(out_dir, exist_ok=True)
        B = int(batch_size or self.cfg.get('batch_size', 8))
        dist = torch.distributed
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        lo, hi, per = shard.shard_range(len(ents), rank, world)
        mine = ents[lo:hi]
        rowref = self.cfg.heads.type == 'RowSharNotReducRef'
        # ColumnProposal2 (configs 2/3/5) and RowSharNotReducRef (config 4: 12 lanes x 144 rows padded into the same [72,144,2] block)
        pipe = TilePipeline(self.net, host_threads=int(self.cfg.get('host_threads', 8)), with_decode_endp=True)
        lanes_all, endp_all = [], []
        counters = np.zeros(12, dtype=np.float64)      # coor TP / segs / DG / gts, endp TP / dets / DG / gts, semantic TP / dets / DG / gts
        buf = self.cfg.validate_buffer if gt_avail else None

        def take(futs):
            for f in futs:
                lanes, kept, pts = f.result()
                k = len(lanes_all)
                lanes_all.append(lanes); endp_all.append(kept)
                if not gt_avail:
                    continue
                gt = datasets.load_eval_gt(mine[k][2], self.cfg, merge_connect_lines=not rowref)
                if 'klane' in measures:                          # runner.py:640-651: cal_coor_measures(coor_label, cls_offset_smooth[:, :])
                    L = int(self.net.heads.num_cls)
                    label = datasets.klane_coor_label(gt['label_raw'], L, int(self.net.heads.row_size))
                    counters[0:4] += metric_utils.cal_coor_measures(label, lanes[:L, :, 0], 'conf', offset_thre=buf)[3:7]
                if 'coor' in measures:                           # :742-768
                    counters[0:4] += metric_utils.cal_coor_measures(gt['lc_coor_raw'], lanes[:, :, 0], 'conf', offset_thre=buf)[3:7]
                if 'endp' in measures:                           # :770-777: output['endp'] = the decode's endpoint map (on CUDA the
                    pred = np.zeros(gt['endp_map'].shape, dtype=np.float32)      # post-processing filters a host COPY of it)
                    if len(pts):
                        pred[pts[:, 0], pts[:, 1]] = 1.
                    counters[4:8] += metric_utils.eval_metric_endp_detector(pred, gt['endp_map'], r_thre=buf * 2)[3:7]
                if 'semantic' in measures:                       # :779-787 on lane_maps['semantic_line'] (renew_semantic_map raster)
                    counters[8:12] += metric_utils.eval_metric_line_segmentor(hostpost.raster_semantic_map(lanes), gt['mask'],
                                                                             bi_seg=False, semantics=2, buff=buf)[3:7]

        # per-tile JSON files (rank 0): written by a few helper threads - the text comes from the C library (lm_lane_json_write, no GIL) -
        # while the next batches run; a single rank starts them as the tiles complete, several ranks after the all-gather
        from concurrent.futures import ThreadPoolExecutor
        writers = ThreadPoolExecutor(max_workers=4) if (write_lane_vertex and rank == 0) else None
        writes = []

        def write_json(name, lanes):
            io_utils.save_lane_seq_2d(io_utils.pack_lane_vertices(np.asarray(lanes, dtype=np.float64)),
                                      os.path.join(out_dir, name + '.json'), with_pervertex_semantics=True)

        def take_and_write(futs):
            k0 = len(lanes_all)
            take(futs)
            if writers is not None and world == 1:
                writes.extend(writers.submit(write_json, mine[k][0], lanes_all[k]) for k in range(k0, len(lanes_all)))

        for proj in self._batches([p for _, p, _ in mine], B):
            take_and_write(pipe.submit(proj))
        take_and_write(pipe.flush())
        if world > 1:
            block = shard.pack_tile_results(lanes_all, endp_all, per, self.device)
            gathered = shard.unpack_gathered(shard.all_gather_results(block))[:len(ents)]     # ONE collective for the whole job
            names = ents
            if gt_avail:                                         # + one 96-byte all-reduce of the counters when a labelled set is scored
                t = torch.from_numpy(counters).to(self.device)
                dist.all_reduce(t)
                counters = t.cpu().numpy()
        else:
            gathered = list(zip(lanes_all, endp_all))
            names = mine
        results = {}
        for (name, _, _), (lanes, endp) in zip(names, gathered):
            results[name] = (lanes, endp)
            if writers is not None and world > 1:
                writes.append(writers.submit(write_json, name, lanes))
        if writers is not None:
            for w in writes:
                w.result()                                       # (an I/O error of any file is raised here)
            writers.shutdown()
        self.counters = counters
        return results

    def infer_lane_coordinate_endpoint_semantics(self, path_ckpt=None, mode_data=None, mode_view=False, gt_avail=True,
                                                 write_lane_vertex=False, eval_coor=True, eval_endp=True, eval_semantic=True,
                                                 *, tiles=None, batch_size=None, work_dirs=None):
        """The reference's entry (runner.py:690-867), same positional / keyword signature.  Returns {image_name: (lanes [72,144,2],
        endpoints [k,2])} (all tiles on every rank); self.metrics holds the nine numbers the reference prints, self.counters the sums."""
        measures = tuple(m for m, on in (('coor', eval_coor), ('endp', eval_endp), ('semantic', eval_semantic)) if on)
        results = self._infer(path_ckpt, mode_data, mode_view, bool(gt_avail), write_lane_vertex, measures, tiles, batch_size, work_dirs)
        c = self.counters
        zero = (0., 0., 0.)
        evaluated = bool(gt_avail) and tiles is None
        coor = _prf(*c[0:4]) if evaluated and eval_coor else zero
        endp = _prf(*c[4:8]) if evaluated and eval_endp else zero
        sem = _prf(*c[8:12]) if evaluated and eval_semantic else zero
        self.metrics = {'coordinate_prec': coor[0], 'coordinate_rec': coor[1], 'coordinate_f1': coor[2],
                        'endpoint_prec': endp[0], 'endpoint_rec': endp[1], 'endpoint_f1': endp[2],
                        'semantic_prec': sem[0], 'semantic_rec': sem[1], 'semantic_f1': sem[2]}
        if not torch.distributed.is_initialized() or torch.distributed.get_rank() == 0:
            for k, v in self.metrics.items():                    # the reference's nine lines (:859-867)
                print(f'{k}={v}')
        return results

    def infer_lane_coordinate(self, path_ckpt=None, mode_view=False, gt_avail=True, write_lane_vertex=False,
                              *, tiles=None, batch_size=None, work_dirs=None):
        """The K-Lane / RowRef entry (runner.py:606-687; config 4): cfg.dataset.test, coordinate measures on cls_offset_smooth[:, :]."""
        results = self._infer(path_ckpt, None, mode_view, bool(gt_avail), write_lane_vertex, ('klane',), tiles, batch_size, work_dirs)
        coor = _prf(*self.counters[0:4]) if (gt_avail and tiles is None) else (0., 0., 0.)
        self.metrics = {'coordinate_prec': coor[0], 'coordinate_rec': coor[1], 'coordinate_f1': coor[2]}
        if not torch.distributed.is_initialized() or torch.distributed.get_rank() == 0:
            for k, v in self.metrics.items():
                print(f'{k}={v}')
        return results

    def infer_las_to_map(self, las_and_params, work_dirs=None, path_ckpt=None, batch_size=None, merge=True):
        """LAS tiles -> map-level 3-D lane lines, every stage of the reference's offline chain on this stack:

          LAS file + tile parameter file (utils/io_utils.py:125-150)
            -> points in HBM (las_io.read_las_raw, shifted by las_read_offset)          [laspy read_las in the reference]
            -> BEV tile on the GPU (lm_bev_raster_batch)                                [external Las2BEV tool]
            -> polylines (TilePipeline) -> <name>.json                                  [Runner :690-867]
            -> LAS-frame polylines (coor_img2pc, elevation from the tile) -> pc/<name>.json / .txt   [coor_img2pc.py]
            -> merged + 0.6 m down-sampled lines -> merged.txt / merged_downsample.txt  [merge_lines.py __main__]

        las_and_params: list of (las_path, param_path) in tile order.  Returns (per-tile dict name -> 3-D lines, merged list).
        Single rank (the merge is sequential over the sorted tiles)."""
        from . import coor_img2pc, las_io, merge_lines as ml
        if path_ckpt:
            self.load_ckpt(path_ckpt)
        out_dir = work_dirs or self.cfg.get('work_dirs', './work_dirs')
        pc_dir = os.path.join(out_dir, 'out_pc_seq_json_dir')
        os.makedirs(pc_dir, exist_ok=True)
        B = int(batch_size or self.cfg.get('batch_size', 8))
        pipe = TilePipeline(self.net)
        H, W = self.cfg.list_img_size_xy[1], self.cfg.list_img_size_xy[0]
        queue, lines3d, pc_files = [], {}, []

        def finish(futs):
            for f in futs:
                name, params, u8 = queue.pop(0)
                lanes, _ = f.result()
                packed = io_utils.pack_lane_vertices(np.asarray(lanes, dtype=np.float64))
                io_utils.save_lane_seq_2d(packed, os.path.join(out_dir, name + '.json'), with_pervertex_semantics=True)
                recs = io_utils.lane_records(packed)
                if len(recs) < 2:                       # load_lane_seq yields nothing for < 2 lines: the reference skips the tile
                    continue
                lens = [r['seq_len'] for r in recs]
                seqs = np.zeros((len(recs), max(lens), 2))
                for i, r in enumerate(recs):
                    seqs[i, :lens[i]] = np.asarray(r['seq'])[:, 0:2]
                pc = coor_img2pc.transform_coordinate_from_img_2_pc(params, seqs, lens, u8)
                lines = [{'seq': pc[i, :lens[i], :], 'seq_len': lens[i], 'init_vertex': pc[i, 0, :], 'end_vertex': pc[i, lens[i] - 1, :]}
                         for i in range(len(recs))]
                io_utils.save_seqs_json(lines, os.path.join(pc_dir, name + '.json'))
                io_utils.save_seqs_txt(lines, os.path.join(pc_dir, name + '.txt'))
                pc_files.append(os.path.join(pc_dir, name + '.json'))
                lines3d[name] = [l['seq'] for l in lines]

        for i in range(0, len(las_and_params), B):
            chunk = las_and_params[i:i + B]
            pts, offs, rpar = [], [0], []
            for las_path, param_path in chunk:
                params = io_utils.load_pc_2_img_transform_paras(param_path)
                p, _ = las_io.read_las_raw(las_path, self.device, shift=params['las_read_offset'])
                pts.append(p)
                offs.append(offs[-1] + p.shape[0])
                rpar.append(io_utils.raster_params_from_file(param_path))
                queue.append([os.path.splitext(os.path.basename(las_path))[0][0:11], params, None])
            tiles, u8 = ops.bev_raster_batch(torch.cat(pts), offs, rpar, H, W, want_u8=True)
            u8_host = u8.cpu().numpy()
            for j in range(len(chunk)):
                queue[len(queue) - len(chunk) + j][2] = u8_host[j]
            finish(pipe.submit(tiles))
        finish(pipe.flush())
        merged = []
        if merge and pc_files:
            merged = ml.merge_lines(pc_files)
            io_utils.save_seqs_list(merged, os.path.join(pc_dir, 'merged.txt'))
            io_utils.save_seqs_list([ml.downsample_seqs(m) for m in merged], os.path.join(pc_dir, 'merged_downsample.txt'))
        return lines3d, merged

    def infer_lane_geometry_segmentation_segmentor(self, path_ckpt=None, mode_view=False, write_lane_vertex=False,
                                                   *, tiles=None, batch_size=None, gt_avail=None):
        """Segmentor config (runner.py:945-1036): {image_name: (seg [1152,1152] u8-valued f32, endpoints [k,2])} over cfg.dataset.test
        (or `tiles=`).  With labels (default: when the split is used) the loop's geometry (bi_seg) and semantic skeleton counters are
        summed and its six lines printed; self.metrics holds them."""
        from . import datasets, metric_utils
        if path_ckpt:
            self.load_ckpt(path_ckpt)
        self._view_notice(mode_view)
        ents = self._entries(None, tiles)
        gt_avail = (tiles is None) if gt_avail is None else (bool(gt_avail) and tiles is None)
        B = int(batch_size or self.cfg.get('batch_size', 8))
        dist = torch.distributed
        world = dist.get_world_size() if dist.is_initialized() else 1
        rank = dist.get_rank() if dist.is_initialized() else 0
        lo, hi, _ = shard.shard_range(len(ents), rank, world)
        mine = ents[lo:hi]
        res = {}
        c = np.zeros(8, dtype=np.float64)          # semantic TP / dets / DG / gts, geometry TP / pts / DG / gts
        for i in range(0, len(mine), B):
            chunk = mine[i:i + B]
            out = self.net({'proj': self._load_batch([p for _, p, _ in chunk])})
            for j, (name, _, ent) in enumerate(chunk):
                seg = out['seg'][j].numpy()
                res[name] = (seg, out['endp_pts'][j])
                if gt_avail:
                    mask = datasets.load_eval_gt(ent, self.cfg, merge_connect_lines=False)['mask']
                    c[0:4] += metric_utils.eval_metric_line_segmentor(seg, mask, bi_seg=False, semantics=2, buff=self.cfg.validate_buffer)[3:7]
                    c[4:8] += metric_utils.eval_metric_line_segmentor(seg, mask, bi_seg=True, semantics=1, buff=self.cfg.validate_buffer)[3:7]
        if world > 1:
            # every rank gets every tile's result (the class maps travel as u8 when that is lossless: 1.3 MB per tile) + one 64-byte
            # all-reduce of the counters
            def small(seg):
                u8 = seg.astype(np.uint8)
                return u8 if np.array_equal(u8.astype(seg.dtype), seg) else seg
            parts = [None] * world
            dist.all_gather_object(parts, {k: (small(v[0]), str(v[0].dtype), v[1]) for k, v in res.items()})
            res = {k: (seg.astype(dt), pts) for part in parts for k, (seg, dt, pts) in part.items()}
            res = {name: res[name] for name, _, _ in ents}
            t = torch.from_numpy(c).to(self.device)
            dist.all_reduce(t)
            c = t.cpu().numpy()
        self.counters = c
        if gt_avail:
            geo, sem = _prf(*c[4:8]), _prf(*c[0:4])
            self.metrics = {'coor_conf_prec': geo[0], 'coor_conf_rec': geo[1], 'coor_conf_f1': geo[2],
                            'sem_conf_prec': sem[0], 'sem_conf_rec': sem[1], 'sem_conf_f1': sem[2]}
            if rank == 0:
                for k, v in self.metrics.items():
                    print(f'{k}={v}')
        return res
