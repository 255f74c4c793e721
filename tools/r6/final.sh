#!/bin/bash
# last evidence pass on the final sources: GPU suite + smoke() with the stamp, PMC traffic (stamped), headline bench line, rocprofv3 kernel stats
R=${GRAFT_REPO_ROOT:?run through gpurun}
O=$R/gpurun_out/${1:-r6_final}
mkdir -p $O
cd $R
tools/r6/suite.sh ${1:-r6_final}
tools/pmc_traffic.sh r6 > $O/pmc_traffic_log.txt 2>&1
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json      # (the bench below reads it: traffic_stale false on these sources)
python bench.py 2>/dev/null | tail -1 > $O/bench_config3_fused.json
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-second-line --streams 1 --conv-detail 2> $O/bench_fused_conv_detail.txt > /dev/null
tools/r6/kernel_stats.sh ${1:-r6_final}/kstats > /dev/null 2>&1
tools/r5/stage_stats.sh gpurun_out/${1:-r6_final}/s1 --streams 1 > /dev/null 2>&1
python -c "
import json
d=json.load(open('$O/bench_config3_fused.json')); print(round(d['value'],1), d['config']['windows_tiles_per_s'], d['roofline']['traffic_stale'], d['roofline']['frac'], (d.get('second_line') or {}).get('value'))
"
