#!/bin/bash
# Re-measure the committed evidence (round 5): bench lines (headline = fused, tiles, rowref, lidar, direct-kernel A/B),
# raster micro-bench, per-layer Winograd table, phase profile, rocprofv3 kernel stats of the default command and of --streams 1
# -> gpurun_out/refresh (copy what is to be judged into profiles/ as r5_*)
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}
O=$R/gpurun_out/refresh
mkdir -p $O
cd $R
python bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/bench_config3_fused.json
LANEMAP_WINO_F44=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3_fused_direct.json
python bench.py --workload tiles --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config2.json
python bench.py --workload rowref --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config4_rowref.json
python bench.py --workload lidar --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config5_lidar.json
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --conv-detail 2> $O/bench_fused_conv_detail.txt > /dev/null
python tools/bench_raster.py 2>/dev/null | tail -1 > $O/raster.json
python tools/r4/bench_wino44.py 8 10 2>/dev/null > $O/wino44_layers_b8.txt
python tools/r4/bench_wino44.py 16 10 2>/dev/null > $O/wino44_layers_b16.txt
[ -f tools/probes/lib_qprof.so ] && LANEMAP_HIP_LIB=$R/tools/probes/lib_qprof.so python tools/r4/qprof.py 8 2>/dev/null > $O/wino44_phase_profile.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o p -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>> $R/gpurun_out/prof_stderr.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s1 -o p -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --streams 1 > /dev/null 2>> $R/gpurun_out/prof_stderr.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_raster -o p -- python3 $R/tools/bench_raster.py > /dev/null 2>> $R/gpurun_out/prof_stderr.log
rm -f $O/prof_*/p_kernel_trace.csv     # large; the stats are what is committed
ls $O $O/prof_default
