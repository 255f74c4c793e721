#!/bin/bash
# Round-6 evidence run (through gpurun): full GPU suite with the sources' stamp, G15 / G17 exact counts under the three 3x3 routes (every
# result logged, pass or fail, with its exit code), bench lines of every workload (the headline with its second line), the input-inclusive
# variants, steady-state stage statistics, per-layer tables (exact and split), raster micro-bench, lateral A/B, PMC traffic.
R=${GRAFT_REPO_ROOT:?run through gpurun}
O=$R/gpurun_out/${1:-r6_evidence}
mkdir -p $O
cd $R
SHA=$(python -c "import bench; print(bench.csrc_sha16())")
( echo "# pytest tests -x -q -m gpu on MI355X; csrc_sha16 = $SHA"; python -m pytest tests -x -q -m gpu --durations=12 ) > $O/gpu_suite.txt 2>&1
tail -3 $O/gpu_suite.txt
rm -f $O/exact_counts.txt
for route in "" "LANEMAP_WINO_F44=0" "LANEMAP_WINO_SPLIT=1"; do
  env $route LANEMAP_PARITY_LOG=$O/exact_counts.txt python -m pytest tests/test_gpu_2_goldens.py -q -m gpu -s -k "stable_golden_g15 or chain_golden_g17" > $O/goldens_route.txt 2>&1
  echo "[${route:-default route}] pytest exit code $? : $(tail -1 $O/goldens_route.txt)" >> $O/exact_counts.txt
  grep "column bin differs" $O/goldens_route.txt >> $O/exact_counts.txt
done
cat $O/exact_counts.txt
python bench.py 2>/dev/null | tail -1 > $O/bench_config3_fused.json
python bench.py --points host --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_fused_hostpoints.json
python bench.py --points las --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_fused_laspoints.json
LANEMAP_WINO_F44=0 python bench.py --steps 40 --no-cpu-baseline --no-second-line 2>/dev/null | tail -1 > $O/bench_config3_fused_direct.json
python bench.py --workload tiles --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config2.json
python bench.py --workload rowref --steps 40 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config4_rowref.json
python bench.py --workload lidar --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config5_lidar.json
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-second-line --streams 1 --conv-detail 2> $O/bench_fused_conv_detail.txt > /dev/null
python tools/bench_raster.py 2>/dev/null | tail -1 > $O/raster.json
python tools/r4/bench_wino44.py 16 10 2>/dev/null > $O/wino44_layers_b16.txt
python tools/r6/bench_wino44_split.py 16 10 2>/dev/null > $O/wino44_split_layers_b16.txt
for i in 1 2; do
  LM_CONV_LATERAL=1 python tools/r6/bench_lateral.py >> $O/lateral_ab.txt 2>/dev/null
  LM_CONV_LATERAL=0 python tools/r6/bench_lateral.py >> $O/lateral_ab.txt 2>/dev/null
done
tools/r5/stage_stats.sh gpurun_out/${1:-r6_evidence}/s1 --streams 1 --no-second-line > /dev/null 2>&1
tools/r5/stage_stats.sh gpurun_out/${1:-r6_evidence}/s2 --streams 2 --no-second-line > /dev/null 2>&1
tools/pmc_traffic.sh r6 > $O/pmc_traffic_log.txt 2>&1
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],1), d['unit'], d['config'].get('windows_tiles_per_s'), (d.get('second_line') or {}).get('value'))
    except Exception as e: print(f, 'ERR', e)
"
