"""Inference runner: the `test_gpu_0.py` / `Runner` entry of the reference, hot-path subset.

  load_config_and_runner(path, gpus)                 <- baseline/engine/runner.py:57-66
  Runner.load_ckpt(path)                             <- :399-401 (strict, 'module.'-prefixed keys accepted)
  Runner.infer_lane_coordinate_endpoint_semantics()  <- :690-867 minus metrics / cv2 overlays: every tile ->
                                                        <work_dirs>/<image_name[0:11]>.json via save_lane_seq_2d
  Runner.infer_lane_geometry_segmentation_segmentor()<- :945-1036 minus overlays (Segmentor config)
  Runner.infer_las_to_map()                          <- the offline chain LAS -> BEV -> polylines -> LAS frame -> merged map
                                                        (read_las, Las2BEV, Runner, coor_img2pc.py, merge_lines.py) in one call
Tiles are PNG files (load_img contract, datasets/laserlane_proposals.py:85-98): decoded on the host by the library's own PNG reader (png_io, zlib on the host thread pool),
converted u8 -> f32/255 on the GPU (lm_tile_ingest_u8).  With torch.distributed initialised, tiles are sharded
over the ranks (lanemapping_amd/shard.py) and rank 0 writes every file after one all-gather per batch.
"""
import glob
import os

import numpy as np
import torch

from . import io_utils, ops, shard
from .boundary import load_config, load_reference_checkpoint
from .pipeline import TilePipeline
from .registry import build_net


def load_config_and_runner(path_config, gpus='0'):
    cfg = load_config(path_config)
    cfg['gpus'] = len(str(gpus).split(','))
    cfg.setdefault('work_dirs', os.path.join(cfg.get('log_dir', './logs'), 'infer'))
    return cfg, Runner(cfg)


class Runner:
    def __init__(self, cfg, device=None):
        self.cfg = cfg
        seed = int(cfg.get('seed', 2021))
        torch.manual_seed(seed)
        np.random.seed(seed)
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
    def infer_lane_coordinate_endpoint_semantics(self, tiles=None, path_ckpt=None, write_lane_vertex=True, batch_size=None,
                                                 work_dirs=None, **_ignored):
        """Returns {image_name: (lanes [72,144,2], endpoints [k,2])} for this rank's tiles (all tiles on rank 0)."""
        if path_ckpt:
            self.load_ckpt(path_ckpt)
        paths = self.list_tiles(tiles if tiles is not None else self.cfg.dataset.test.data_root)
        out_dir = work_dirs or self.cfg.get('work_dirs', './work_dirs')
        os.makedirs(out_dir, exist_ok=True)
        B = int(batch_size or self.cfg.get('batch_size', 8))
        world = torch.distributed.get_world_size() if torch.distributed.is_initialized() else 1
        rank = torch.distributed.get_rank() if torch.distributed.is_initialized() else 0
        lo, hi, per = shard.shard_range(len(paths), rank, world)
        mine = paths[lo:hi]
        results = {}
        lanes_all, endp_all = [], []
        # ColumnProposal2 (configs 2/3/5) and RowSharNotReducRef (config 4: 12 lanes x 144 rows padded into the same [72,144,2] block)
        pipe = TilePipeline(self.net, host_threads=int(self.cfg.get('host_threads', 8)))
        for proj in self._batches(mine, B):
            for f in pipe.submit(proj):
                lanes_all.append(f.result()[0]); endp_all.append(f.result()[1])
        for f in pipe.flush():
            lanes_all.append(f.result()[0]); endp_all.append(f.result()[1])
        if world > 1:
            block = shard.pack_tile_results(lanes_all, endp_all, per, self.device)
            gathered = shard.unpack_gathered(shard.all_gather_results(block))         # ONE collective for the whole job
            names = paths
        else:
            gathered = list(zip(lanes_all, endp_all))
            names = mine
        for p, (lanes, endp) in zip(names, gathered):
            name = os.path.splitext(os.path.basename(p))[0][0:11]
            results[name] = (lanes, endp)
            if write_lane_vertex and rank == 0:
                io_utils.save_lane_seq_2d(io_utils.pack_lane_vertices(np.asarray(lanes, dtype=np.float64)),
                                          os.path.join(out_dir, name + '.json'), with_pervertex_semantics=True)
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

    def infer_lane_geometry_segmentation_segmentor(self, tiles=None, path_ckpt=None, batch_size=None, **_ignored):
        """Segmentor config: {image_name: (seg [1152,1152] u8-valued f32, endpoints [k,2])}."""
        if path_ckpt:
            self.load_ckpt(path_ckpt)
        paths = self.list_tiles(tiles if tiles is not None else self.cfg.dataset.test.data_root)
        B = int(batch_size or self.cfg.get('batch_size', 8))
        res = {}
        for i in range(0, len(paths), B):
            out = self.net({'proj': self._load_batch(paths[i:i + B])})
            for j, p in enumerate(paths[i:i + B]):
                res[os.path.splitext(os.path.basename(p))[0][0:11]] = (out['seg'][j].numpy(), out['endp_pts'][j])
        return res
