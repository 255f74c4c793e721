#!/bin/bash
# rocprofv3 kernel durations of the rasteriser: config-3 cloud vs uniform points
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}
for mode in cloud uniform; do
  if [ $mode = uniform ]; then export RASTER_UNIFORM=1; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/rast_$mode -o r -- python3 $R/tools/bench_raster.py > /dev/null 2>> $R/gpurun_out/prof_stderr.log
  echo $mode; grep raster_ $R/gpurun_out/rast_$mode/r_kernel_stats.csv | awk -F'",' '{print $1}' | cut -c1-60 | paste - <(grep raster_ $R/gpurun_out/rast_$mode/r_kernel_stats.csv | awk -F',' '{print $(NF-6), $(NF-5), $(NF-4)}')
done
