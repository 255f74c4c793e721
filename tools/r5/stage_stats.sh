#!/bin/bash
# tools/r5/stage_stats.sh OUTDIR [bench args]: steady-state per-kernel / per-stage statistics of bench.py (default --streams 1) -> $OUTDIR/stage_stats.txt
# (kernel by kernel, --no-graphs: a graph replay has one launch record for all its kernels, which would all land outside the stage ranges)
R=${GRAFT_REPO_ROOT:?run through gpurun}
OUT=$R/$1; shift
ARGS=${*:---streams 1}
STEPS=6
mkdir -p $OUT
export LANEMAP_ROCTX=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --marker-trace --hip-runtime-trace --output-format csv -d $OUT/trace -o p -- python3 $R/bench.py --steps $STEPS --warmup 2 --no-cpu-baseline --no-second-line --no-graphs $ARGS > $OUT/bench.json 2> $OUT/bench.err
cd $R
python3 tools/r5/stage_stats.py $OUT/trace $STEPS > $OUT/stage_stats.txt 2> $OUT/stage_stats.err
rm -rf $OUT/trace
cat $OUT/stage_stats.txt $OUT/stage_stats.err | head -90
