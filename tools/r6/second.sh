#!/bin/bash
# round 6, second GPU call: lateral kernel A/B (bits + time), two-id entry test, input-inclusive bench variants
R=${GRAFT_REPO_ROOT:?run through gpurun}
O=$R/gpurun_out/${1:-r6_second}
mkdir -p $O
cd $R
python -m pytest tests/test_gpu_1_kernels.py -x -q -m gpu -k "lateral or conv1x1 or resup or upsample_add" > $O/lateral_tests.txt 2>&1; tail -5 $O/lateral_tests.txt
LM_CONV_LATERAL=1 python tools/r6/bench_lateral.py > $O/lateral_on.json 2>/dev/null
LM_CONV_LATERAL=0 python tools/r6/bench_lateral.py > $O/lateral_off.json 2>/dev/null
LM_CONV_LATERAL=1 python tools/r6/bench_lateral.py >> $O/lateral_on.json 2>/dev/null
LM_CONV_LATERAL=0 python tools/r6/bench_lateral.py >> $O/lateral_off.json 2>/dev/null
cat $O/lateral_on.json $O/lateral_off.json
python -m pytest tests/test_gpu_3_configs.py -x -q -m gpu -k "two_gpu_ids" > $O/two_ids.txt 2>&1; tail -5 $O/two_ids.txt
python -m pytest tests/test_gpu_9_bench.py -x -q -m gpu -k "input_inclusive" > $O/points_tests.txt 2>&1; tail -15 $O/points_tests.txt
python bench.py --steps 60 --no-cpu-baseline --points host 2>$O/bench_host.err | tail -1 > $O/bench_fused_hostpoints.json
python bench.py --steps 60 --no-cpu-baseline --points las 2>$O/bench_las.err | tail -1 > $O/bench_fused_laspoints.json
python bench.py --steps 60 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_fused.json
python -c "
import json
for f in ('bench_fused','bench_fused_hostpoints','bench_fused_laspoints'):
    try:
        d=json.load(open('$O/'+f+'.json')); print(f, round(d['value'],1), d['config'].get('point_feed'), d['config']['windows_tiles_per_s'])
    except Exception as e: print(f,'ERR',e)
"
tail -3 $O/bench_host.err $O/bench_las.err
