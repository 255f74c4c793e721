#!/bin/bash
# Round-5 evidence run (through gpurun): full GPU suite with the sources' stamp, bench lines of every workload, steady-state stage statistics,
# exact-count log of G15 / G17 for the default route and LANEMAP_WINO_F44=0, per-layer Winograd table, raster micro-bench, PMC traffic.
R=${GRAFT_REPO_ROOT:?run through gpurun}
O=$R/gpurun_out/${1:-r5_evidence}
mkdir -p $O
cd $R
SHA=$(python -c "import bench; print(bench.csrc_sha16())")
( echo "# pytest tests -x -q -m gpu on MI355X; csrc_sha16 = $SHA"; python -m pytest tests -x -q -m gpu --durations=10 ) > $O/gpu_suite.txt 2>&1
tail -3 $O/gpu_suite.txt
rm -f $O/exact_counts.txt
LANEMAP_PARITY_LOG=$O/exact_counts.txt python -m pytest tests/test_gpu_2_goldens.py -q -m gpu -k "stable_golden_g15 or chain_golden_g17" > /dev/null 2>&1
LANEMAP_WINO_F44=0 LANEMAP_PARITY_LOG=$O/exact_counts.txt python -m pytest tests/test_gpu_2_goldens.py -q -m gpu -k "stable_golden_g15 or chain_golden_g17" > /dev/null 2>&1
cat $O/exact_counts.txt
python bench.py 2>/dev/null | tail -1 > $O/bench_config3_fused.json
LANEMAP_WINO_F44=0 python bench.py --steps 40 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3_fused_direct.json
python bench.py --workload tiles --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config2.json
python bench.py --workload rowref --steps 40 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config4_rowref.json
python bench.py --workload lidar --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config5_lidar.json
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --streams 1 --conv-detail 2> $O/bench_fused_conv_detail.txt > /dev/null
python tools/bench_raster.py 2>/dev/null | tail -1 > $O/raster.json
python tools/r4/bench_wino44.py 16 10 2>/dev/null > $O/wino44_layers_b16.txt
tools/r5/stage_stats.sh gpurun_out/${1:-r5_evidence}/s1 --streams 1 > /dev/null 2>&1
tools/r5/stage_stats.sh gpurun_out/${1:-r5_evidence}/s2 --streams 2 > /dev/null 2>&1
tools/pmc_traffic.sh r5 > $O/pmc_traffic_log.txt 2>&1
python -c "
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.load(open(f)); print(f.split('/')[-1], round(d['value'],1), d['unit'], d['config'].get('windows_tiles_per_s'))
    except Exception as e: print(f, 'ERR', e)
"
