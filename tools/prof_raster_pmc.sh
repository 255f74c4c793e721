#!/bin/bash
# HBM traffic of the rasteriser kernels from PMC counters (separate passes, no tracing): FETCH_SIZE, WRITE_SIZE
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $R/gpurun_out/pmc_raster/$c -o p -- python3 $R/tools/bench_raster.py > /dev/null 2>> $R/gpurun_out/prof_stderr.log
done
python3 - <<PY
import csv, glob, collections
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    f = glob.glob('$R/gpurun_out/pmc_raster/%s/*counter_collection.csv' % c)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        agg[(r['Kernel_Name'][:60], r['Counter_Name'])].append(float(r['Counter_Value']))
    for (k, n), v in agg.items():
        if 'raster' in k or 'fill' in k.lower():
            print(c, k, n, 'launches', len(v), 'mean per launch', sum(v) / len(v))
PY
