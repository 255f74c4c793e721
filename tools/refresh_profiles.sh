#!/bin/bash
# Re-measure the committed evidence: bench lines (headline = fused, tiles, rowref), raster micro-bench, rocprofv3 kernel stats of the
# default command and of --streams 1 -> gpurun_out/refresh (copy what is to be judged into profiles/ as r3_*)
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}
O=$R/gpurun_out/refresh
mkdir -p $O
cd $R
python bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/bench_config3_fused.json
python bench.py --workload tiles --steps 10 --warmup 3 --no-cpu-baseline --no-second-line 2>/dev/null | tail -1 > $O/bench_config2.json
python bench.py --workload rowref --steps 5 --warmup 2 --no-cpu-baseline --no-second-line 2>/dev/null | tail -1 > $O/bench_config4_rowref.json
python tools/bench_raster.py 2>/dev/null | tail -1 > $O/raster.json
python tools/bench_lidar.py 8 > $O/lidar_config5.txt 2>/dev/null
python bench.py --workload lidar --steps 5 --warmup 2 --no-cpu-baseline --no-second-line 2>/dev/null | tail -1 > $O/bench_config5_lidar.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o p -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-second-line > /dev/null 2>> $R/gpurun_out/prof_stderr.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s1 -o p -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-second-line --streams 1 > /dev/null 2>> $R/gpurun_out/prof_stderr.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_raster -o p -- python3 $R/tools/bench_raster.py > /dev/null 2>> $R/gpurun_out/prof_stderr.log
# the second line's kernels (split-precision Winograd GEMMs), single stream so that durations add up; its per-layer table, phase ticks, probe
LANEMAP_WINO_BF16X3=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_split_s1 -o p -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-second-line --streams 1 > /dev/null 2>> $R/gpurun_out/prof_stderr.log
cd $R
python tools/r3/bench_split.py 8 2>/dev/null > $O/split_layers.txt
[ -f tools/probes/libvar_iprof.so ] && LANEMAP_HIP_LIB=$R/tools/probes/libvar_iprof.so python tools/r3/bench_split.py 8 2>/dev/null | grep iprof > $O/rows_iprof.txt
[ -x tools/probes/split_probe ] && tools/probes/split_probe > $O/split_probe.txt
rm -f $O/prof_*/p_kernel_trace.csv     # large; the stats are what is committed
ls $O $O/prof_default
