#!/bin/bash
# HBM traffic of the bench's dominant kernels from the PMC counters, collected as MI355X_MICROARCH.md prescribes: FETCH_SIZE and
# WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (TCC has 4 slots: FETCH_SIZE takes 3, WRITE_SIZE 2), FETCH_SIZE doubled (gfx950
# reports half of a wide streaming read), units of 1 KiB per count as rocprofv3 reports them for these derived counters.
# Writes profiles/pmc_traffic.json (read by bench.py) + the per-kernel table under profiles/.   usage: tools/pmc_traffic.sh [tag]
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}
TAG=${1:-r4}
O=$R/gpurun_out/pmc_traffic
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for wl in fused tiles; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d $O/${wl}_$c -o p -- python3 $R/bench.py --workload $wl --steps 2 --warmup 1 --streams 1 --no-graphs --no-cpu-baseline --no-second-line > /dev/null 2>> $R/gpurun_out/prof_stderr.log
  done
done
python3 - <<PY
import csv, glob, json, collections, sys
sys.path.insert(0, '$R')
from bench import csrc_sha16, kernel_sha16          # stamp of the kernel sources these counters belong to (bench.py prints traffic_stale when it differs)
out = {}
table = []
for wl in ('fused', 'tiles'):
    per = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    for c in ('FETCH_SIZE', 'WRITE_SIZE'):
        f = glob.glob('$O/%s_%s/*counter_collection.csv' % (wl, c))[0]
        for r in csv.DictReader(open(f)):
            n = r['Kernel_Name']
            k = ('wino44' if 'wino44_kernel' in n else 'wino_rows' if 'wino_rows' in n else 'wino_gemm' if 'wino_gemm' in n else 'wino_implicit' if ('wino_implicit' in n or 'wino_dual' in n or 'wino_pipe' in n) else 'wino_input' if 'wino_input' in n else
                 'conv_mfma' if ('conv_mfma' in n or 'lateral_mfma' in n) else 'raster_partition' if 'raster_partition' in n else 'raster_band' if 'raster_band' in n else None)
            if k is None: continue
            per[k][c] += float(r['Counter_Value'])
            if c == 'FETCH_SIZE': launches[k] += 1
    # batches in the run: priming + warmup + steps = 4 (each runs the whole net once on the full batch)
    steps = 4.0
    KB = 1024.0
    def bytes_of(k):
        return (2.0 * per[k]['FETCH_SIZE'] + per[k]['WRITE_SIZE']) * KB / steps
    mfma = sum(bytes_of(k) for k in ('wino44', 'wino_rows', 'wino_gemm', 'wino_implicit', 'wino_input', 'conv_mfma') if k in per)
    rast = sum(bytes_of(k) for k in ('raster_partition', 'raster_band') if k in per)
    out[wl] = {'mfma_bytes_per_step': mfma, 'raster_bytes_per_step': rast if rast else None, 'csrc_sha16': csrc_sha16(), 'kernel_sha16': kernel_sha16(),
               'source': 'profiles/${TAG}_pmc_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes on bench.py --workload %s --streams 1; FETCH x 2 + WRITE, KiB units)' % wl}
    for k in per:
        table.append('%-6s %-18s launches/step %6.1f  fetch(x2) %10.1f MB/step  write %10.1f MB/step' % (wl, k, launches[k] / steps, 2 * per[k]['FETCH_SIZE'] * KB / steps / 1e6, per[k]['WRITE_SIZE'] * KB / steps / 1e6))
json.dump(out, open('$R/gpurun_out/pmc_traffic.json', 'w'), indent=1)
open('$R/gpurun_out/${TAG}_pmc_traffic.txt', 'w').write('\n'.join(table) + '\n')
print('\n'.join(table)); print(json.dumps(out))
PY
