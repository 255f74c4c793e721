#!/bin/bash
R=${GRAFT_REPO_ROOT:?run through gpurun}
O=$R/gpurun_out/${1:-r6_sweep}
mkdir -p $O
cd $R
for rep in 1 2; do
for s in 2 1 3 4; do
  python bench.py --steps 60 --no-cpu-baseline --no-second-line --streams $s 2>/dev/null | tail -1 > $O/s${s}_$rep.json
done
done
python -c "
import json,glob
for f in sorted(glob.glob('$O/s*.json')):
    d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],1), d['config']['windows_tiles_per_s']['median'] if d['config']['windows_tiles_per_s'] else None)
"
