#!/bin/bash
# file-to-file path (Runner: PNG tiles -> polylines -> JSON) + the PNG / DEFLATE reader timings on the GPU box's host cores
R=${GRAFT_REPO_ROOT:?run through gpurun}
O=$R/gpurun_out/${1:-r6_runner_files}
mkdir -p $O
cd $R
{
  nproc
  python tools/bench_runner.py 64
  python tools/bench_runner.py 256
  python tools/bench_png.py
} > $O/runner_files.txt 2>&1
python -m pytest tests/test_boundary_cpu.py -q -k "png or zlib" > $O/png_tests.txt 2>&1
tail -3 $O/png_tests.txt
cat $O/runner_files.txt
