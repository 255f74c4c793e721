"""`GPUS_EN = '0,1,...'`: the reference's way to use several GPUs, made real.

The reference's entry script masks the devices (`os.environ["CUDA_VISIBLE_DEVICES"] = GPUS_EN`, test_gpu_0.py:7-9), passes the same string
to `load_config_and_runner(path_config, GPUS_EN)` (baseline/engine/runner.py:57-66: `cfg.gpus = len(gpus.split(','))`) and `Runner`
wraps the net in `DataParallel(device_ids=range(cfg.gpus))` (:103-104): ONE process scatters every batch over the listed GPUs.  A
DataParallel scatter is the wrong shape for this stack (one process drives one GPU: the C library launches on the current HIP device),
so a multi-id string gives a `MultiGpuRunner` instead of a `Runner`:

  * the parent never touches a GPU: it holds the config and a CPU copy of the network (`load_ckpt` loads into it, strict), so that
    whatever the caller did to `runner.net` / `runner.cfg` before the call travels to the ranks;
  * every `infer_*` call starts ONE FRESH CHILD PROCESS PER LISTED GPU (`python -m lanemapping_amd.runner_ranks <job>`; nothing is
    exec'ed from a process that has initialised HIP), rank r pinned to its GPU, `torch.distributed` over RCCL (`nccl`), tiles
    block-sharded by `shard.py`, ONE all-gather of the per-tile result blocks, rank 0 writing every file and printing the reference's
    P / R / F1 lines - the code path `torchrun` + `Runner` takes, which `test_runner_two_ranks_byte_identical` holds byte-identical to
    the single-rank run;
  * the call returns what the single-GPU call returns (the results of ALL tiles) and sets `runner.metrics` / `runner.counters`.

Which devices: the ids are indices into what the process can see.  When the caller masked the devices the reference's way
(`CUDA_VISIBLE_DEVICES` / `HIP_VISIBLE_DEVICES` == the gpus string) the ranks take the visible devices 0..n-1, exactly DataParallel's
`range(cfg.gpus)`; without a mask rank r takes device ids[r].  More ids than visible devices raises at construction with the
equivalent `torch.distributed.run` command.  LANEMAP_TEST_DEVICE=<d> (tests on a 1-GPU box) puts every rank on device d over `gloo`
(RCCL refuses two ranks on one device).  LANEMAP_RANKS_TIMEOUT=<seconds> bounds a call (default: none - a tile set may take hours);
when it expires, or when one rank fails, the remaining ranks are ended by PID and the call raises with every rank's stderr.
"""
import os
import pickle
import subprocess
import sys
import tempfile
import time

REPO_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_gpus(gpus):
    """'0' / '0,1,2' / 3 -> [ids]; anything else (empty fields, non-integers, negative or repeated ids) raises ValueError."""
    fields = [f.strip() for f in str(gpus).split(',')]
    try:
        ids = [int(f) for f in fields]
    except ValueError:
        raise ValueError(f'gpus={gpus!r}: expected a comma-separated list of GPU ids such as "0" or "0,1,2,3"') from None
    if any(i < 0 for i in ids) or len(set(ids)) != len(ids):
        raise ValueError(f'gpus={gpus!r}: GPU ids must be distinct and non-negative')
    return ids


def _plain(v):
    """ConfigDict tree -> plain dict / list / tuple tree (what travels to the ranks)."""
    if isinstance(v, dict):
        return {k: _plain(x) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return type(v)(_plain(x) for x in v)
    return v


def torchrun_command(n):
    return (f'python -m torch.distributed.run --nnodes=1 --nproc-per-node {n} --master-addr 127.0.0.1 --master-port <port> <script>.py   '
            f'# the script calls torch.distributed.init_process_group("nccl") and load_config_and_runner(path, "0"): '
            f'Runner pins rank r to cuda:$LOCAL_RANK and shards the tiles (lanemapping_amd/shard.py)')


def rank_devices(ids):
    """Device index of every rank + the process-group backend (module docstring, 'Which devices')."""
    test_dev = os.environ.get('LANEMAP_TEST_DEVICE')
    if test_dev is not None:
        return [int(test_dev)] * len(ids), 'gloo'
    mask = os.environ.get('HIP_VISIBLE_DEVICES', os.environ.get('CUDA_VISIBLE_DEVICES'))
    if mask is not None and [f.strip() for f in mask.split(',')] == [str(i) for i in ids]:
        return list(range(len(ids))), 'nccl'
    return list(ids), 'nccl'


class MultiGpuRunner:
    """What `load_config_and_runner(path, '0,1,...')` returns: the `Runner` interface of the reference's entry script, every
    inference call fanned out over one fresh process per GPU (module docstring)."""

    def __init__(self, cfg, gpu_ids):
        import torch
        from .registry import build_net
        self.cfg = cfg
        self.gpu_ids = list(gpu_ids)
        self.devices, self.backend = rank_devices(self.gpu_ids)
        if 'LANEMAP_TEST_DEVICE' not in os.environ:
            seen = torch.cuda.device_count()            # (counting devices does not initialise HIP)
            if max(self.devices) >= seen:
                raise RuntimeError(
                    f'gpus={",".join(map(str, self.gpu_ids))!r} names {len(self.gpu_ids)} GPUs (device indices {self.devices}) but this '
                    f'process sees {seen}; list visible devices, or start the ranks yourself: ' + torchrun_command(len(self.gpu_ids)))
        torch.manual_seed(int(cfg.get('seed', 2021)))
        self.net = build_net(cfg).eval()                # CPU copy: load_ckpt / caller edits land here and travel to the ranks
        self.metrics, self.counters = {}, None
        self.timeout = float(os.environ.get('LANEMAP_RANKS_TIMEOUT', 0)) or None

    def load_ckpt(self, path_ckpt):
        from .boundary import load_reference_checkpoint
        return load_reference_checkpoint(self.net, path_ckpt, strict=True)

    # ------------------------------------------------------------------------------------------------ the fan-out
    def _launch(self, method, kwargs):
        import socket
        import torch
        with tempfile.TemporaryDirectory(prefix='lanemap_ranks_') as tmp:
            job = {'cfg': _plain(self.cfg), 'method': method, 'kwargs': _plain(kwargs), 'devices': self.devices,
                   'backend': self.backend, 'state': os.path.join(tmp, 'state.pt'), 'result': os.path.join(tmp, 'result.pkl')}
            torch.save(self.net.state_dict(), job['state'])
            with open(os.path.join(tmp, 'job.pkl'), 'wb') as f:
                pickle.dump(job, f)
            s = socket.socket()
            s.bind(('127.0.0.1', 0))
            port = s.getsockname()[1]
            s.close()
            n = len(self.devices)
            procs, logs = [], []
            for r in range(n):
                env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                           MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                           PYTHONPATH=REPO_ROOT + os.pathsep + os.environ.get('PYTHONPATH', ''))
                env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
                log = open(os.path.join(tmp, f'rank{r}.err'), 'w+')
                logs.append(log)
                # rank 0 prints the reference's lines through the caller's stdout; every rank's stderr is kept for the error message
                procs.append(subprocess.Popen([sys.executable, '-m', 'lanemapping_amd.runner_ranks', os.path.join(tmp, 'job.pkl')],
                                              env=env, stdout=None if r == 0 else subprocess.DEVNULL, stderr=log))
            codes = self._wait(procs)
            if any(codes):
                tails = []
                for r, log in enumerate(logs):
                    log.seek(0)
                    tails.append(f'--- rank {r} (GPU {self.gpu_ids[r]}, exit code {codes[r]}) ---\n' + log.read()[-3000:])
                raise RuntimeError(f'{method}: {sum(1 for c in codes if c)} of {n} GPU ranks failed\n' + '\n'.join(tails))
            for log in logs:
                log.close()
            with open(job['result'], 'rb') as f:
                out = pickle.load(f)
        self.metrics, self.counters = out['metrics'], out['counters']
        return out['results']

    def _wait(self, procs):
        """Exit codes of the ranks; when one rank fails the others (blocked in a collective it will never join) are ended, by PID."""
        t0 = time.time()
        while True:
            codes = [p.poll() for p in procs]
            if all(c is not None for c in codes):
                return codes
            failed = any(c not in (None, 0) for c in codes)
            if failed or (self.timeout and time.time() - t0 > self.timeout):
                time.sleep(2.0 if failed else 0.0)              # (let the peers fail by themselves first: their own message is better)
                for p in procs:
                    if p.poll() is None:
                        p.terminate()
                for p in procs:
                    try:
                        p.wait(timeout=20)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        p.wait()
                return [p.returncode if p.returncode is not None else -9 for p in procs]
            time.sleep(0.05)

    # ------------------------------------------------------------------------------------------------ the reference's entries
    def infer_lane_coordinate_endpoint_semantics(self, path_ckpt=None, mode_data=None, mode_view=False, gt_avail=True,
                                                 write_lane_vertex=False, eval_coor=True, eval_endp=True, eval_semantic=True,
                                                 *, tiles=None, batch_size=None, work_dirs=None):
        """baseline/engine/runner.py:690-867 over every listed GPU; returns {image_name: (lanes, endpoints)} of ALL tiles."""
        if path_ckpt:
            self.load_ckpt(path_ckpt)
        return self._launch('infer_lane_coordinate_endpoint_semantics', dict(
            mode_data=mode_data, mode_view=mode_view, gt_avail=gt_avail, write_lane_vertex=write_lane_vertex, eval_coor=eval_coor,
            eval_endp=eval_endp, eval_semantic=eval_semantic, tiles=tiles, batch_size=batch_size, work_dirs=work_dirs))

    def infer_lane_coordinate(self, path_ckpt=None, mode_view=False, gt_avail=True, write_lane_vertex=False,
                              *, tiles=None, batch_size=None, work_dirs=None):
        """baseline/engine/runner.py:606-687 (K-Lane / RowRef entry) over every listed GPU."""
        if path_ckpt:
            self.load_ckpt(path_ckpt)
        return self._launch('infer_lane_coordinate', dict(mode_view=mode_view, gt_avail=gt_avail, write_lane_vertex=write_lane_vertex,
                                                          tiles=tiles, batch_size=batch_size, work_dirs=work_dirs))

    def infer_lane_geometry_segmentation_segmentor(self, path_ckpt=None, mode_view=False, write_lane_vertex=False,
                                                   *, tiles=None, batch_size=None, gt_avail=None):
        """baseline/engine/runner.py:945-1036 (Segmentor entry) over every listed GPU."""
        if path_ckpt:
            self.load_ckpt(path_ckpt)
        return self._launch('infer_lane_geometry_segmentation_segmentor', dict(
            mode_view=mode_view, write_lane_vertex=write_lane_vertex, tiles=tiles, batch_size=batch_size, gt_avail=gt_avail))

    def infer_las_to_map(self, *a, **k):
        raise NotImplementedError('infer_las_to_map is a single-GPU chain (the cross-tile merge is sequential over the sorted tiles): '
                                  'use load_config_and_runner(path, "<one id>")')


def _rank_main(job_path):
    """One rank of a MultiGpuRunner call (a fresh process: the GPU is first touched here)."""
    import torch
    import torch.distributed as dist
    from .config import Config
    from .runner import Runner
    with open(job_path, 'rb') as f:
        job = pickle.load(f)
    rank = int(os.environ['RANK'])
    dev = int(job['devices'][rank])
    if not torch.cuda.is_available() or dev >= torch.cuda.device_count():
        raise RuntimeError(f'rank {rank}: GPU index {dev} is not visible ({torch.cuda.device_count()} devices)')
    torch.cuda.set_device(dev)
    dist.init_process_group(job['backend'])
    try:
        cfg = Config(job['cfg'])
        runner = Runner(cfg, device=torch.device('cuda', dev))
        runner.net.load_state_dict(torch.load(job['state'], map_location='cpu'), strict=True)
        kwargs = dict(job['kwargs'])
        if isinstance(kwargs.get('mode_data'), dict):
            kwargs['mode_data'] = Config(kwargs['mode_data'])
        results = getattr(runner, job['method'])(**kwargs)
        if rank == 0:
            with open(job['result'], 'wb') as f:
                pickle.dump({'results': results, 'metrics': getattr(runner, 'metrics', {}), 'counters': runner.counters}, f)
        dist.barrier()
    finally:
        dist.destroy_process_group()


if __name__ == '__main__':
    _rank_main(sys.argv[1])
