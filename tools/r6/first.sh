#!/bin/bash
# round 6, first GPU call: the new two-id entry test, the G17 study under both 3x3 routes, a baseline of the headline
R=${GRAFT_REPO_ROOT:?run through gpurun}
O=$R/gpurun_out/${1:-r6_first}
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_3_configs.py -x -q -m gpu -k "two_gpu_ids or two_ranks_byte" > $O/two_ids.txt 2>&1; tail -5 $O/two_ids.txt
python tests/study_g17_direct.py > $O/g17_f44.txt 2>&1
LANEMAP_WINO_F44=0 python tests/study_g17_direct.py > $O/g17_direct.txt 2>&1
LANEMAP_WINO_F44=0 python -m pytest tests/test_gpu_2_goldens.py -q -m gpu -k "chain_golden_g17" > $O/g17_direct_test.txt 2>&1
tail -30 $O/g17_direct_test.txt
cat $O/g17_direct.txt
python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_fused.json
python bench.py --steps 6 --warmup 2 --no-cpu-baseline --streams 1 --conv-detail 2> $O/conv_detail.txt > /dev/null
head -30 $O/conv_detail.txt
python -c "import json; d=json.load(open('$O/bench_fused.json')); print(d['value'], d['config']['windows_tiles_per_s'])"
