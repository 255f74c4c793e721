#!/bin/bash
# round 6, third GPU call: lateral kernel v2 (bits + time), goldens with the margin-aware column bins under both 3x3 routes
R=${GRAFT_REPO_ROOT:?run through gpurun}
O=$R/gpurun_out/${1:-r6_third}
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_1_kernels.py -x -q -m gpu -k "lateral or conv1x1 or resup or upsample_add" > $O/lateral_tests.txt 2>&1; tail -5 $O/lateral_tests.txt
for i in 1 2; do
LM_CONV_LATERAL=1 python tools/r6/bench_lateral.py >> $O/lateral_on.json 2>/dev/null
LM_CONV_LATERAL=0 python tools/r6/bench_lateral.py >> $O/lateral_off.json 2>/dev/null
done
cat $O/lateral_on.json $O/lateral_off.json
rm -f $O/exact_counts.txt
LANEMAP_PARITY_LOG=$O/exact_counts.txt python -m pytest tests/test_gpu_2_goldens.py -q -m gpu -k "stable_golden_g15 or chain_golden_g17" > $O/goldens_f44.txt 2>&1; tail -3 $O/goldens_f44.txt
LANEMAP_WINO_F44=0 LANEMAP_PARITY_LOG=$O/exact_counts.txt python -m pytest tests/test_gpu_2_goldens.py -q -m gpu -s -k "stable_golden_g15 or chain_golden_g17" > $O/goldens_direct.txt 2>&1; echo "direct route exit code $?" >> $O/exact_counts.txt; tail -3 $O/goldens_direct.txt
cat $O/exact_counts.txt
grep "column bin differs" $O/goldens_direct.txt
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --streams 1 --conv-detail 2>&1 >/dev/null | grep "k1x1" 
