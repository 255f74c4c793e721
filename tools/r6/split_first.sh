#!/bin/bash
R=${GRAFT_REPO_ROOT:?run through gpurun}
O=$R/gpurun_out/${1:-r6_split1}
mkdir -p $O
cd $R
timeout 900 python -m pytest tests/test_gpu_1_kernels.py -x -q -m gpu -s -k "split_second_line" > $O/split_test.txt 2>&1; tail -25 $O/split_test.txt
timeout 600 python tools/r6/bench_wino44_split.py 16 10 > $O/wino44_split_layers_b16.txt 2>&1; cat $O/wino44_split_layers_b16.txt
timeout 900 python -m pytest tests/test_gpu_1_kernels.py -x -q -m gpu -k "winograd44" > $O/wino44_tests.txt 2>&1; tail -3 $O/wino44_tests.txt
