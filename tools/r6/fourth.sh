#!/bin/bash
R=${GRAFT_REPO_ROOT:?run through gpurun}
O=$R/gpurun_out/${1:-r6_fourth}
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_1_kernels.py -x -q -m gpu -k "lateral or conv1x1 or resup or upsample_add" > $O/lateral_tests.txt 2>&1; tail -5 $O/lateral_tests.txt
for i in 1 2; do
LM_CONV_LATERAL=1 python tools/r6/bench_lateral.py >> $O/lateral_on.json 2>/dev/null
LM_CONV_LATERAL=0 python tools/r6/bench_lateral.py >> $O/lateral_off.json 2>/dev/null
done
cat $O/lateral_on.json $O/lateral_off.json
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --streams 1 --conv-detail 2>&1 >/dev/null | grep "k1x1" 
python -m pytest tests/test_gpu_3_configs.py -x -q -m gpu -k "two_gpu_ids" > $O/two_ids.txt 2>&1; tail -5 $O/two_ids.txt
