#!/bin/bash
# Re-measure the committed evidence: bench lines + rocprofv3 kernel stats (default command and --streams 1) -> gpurun_out/refresh
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}
O=$R/gpurun_out/refresh
mkdir -p $O
cd $R
python bench.py --steps 10 --warmup 3 2>/dev/null | tail -1 > $O/bench_config2.json
python bench.py --workload fused --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_config3_fused.json
python tools/bench_raster.py 2>/dev/null | tail -1 > $O/raster.json
python tools/bench_lidar.py 8 > $O/lidar.json 2> $O/lidar_layers.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -o p -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>> $R/gpurun_out/prof_stderr.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_s1 -o p -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --streams 1 > /dev/null 2>> $R/gpurun_out/prof_stderr.log
rm -f $O/prof_*/p_kernel_trace.csv     # large; the stats are what is committed
ls $O $O/prof_default
