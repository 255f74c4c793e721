#!/bin/bash
R=${GRAFT_REPO_ROOT:?run through gpurun}
O=$R/gpurun_out/${1:-r6_split7}
mkdir -p $O
cd $R
rm -f $O/exact_counts.txt
LANEMAP_WINO_SPLIT=1 LANEMAP_PARITY_LOG=$O/exact_counts.txt timeout 1200 python -m pytest tests/test_gpu_2_goldens.py -q -m gpu -k "golden_g10 or stable_golden_g15 or chain_golden_g17 or graph_replay or full_tiles_other or inside_full_batches" > $O/goldens_split.txt 2>&1; tail -4 $O/goldens_split.txt
cat $O/exact_counts.txt
python tests/study_numerics_e2e.py > $O/numerics_e2e.txt 2>/dev/null
LANEMAP_WINO_SPLIT=1 python tests/study_numerics_e2e.py >> $O/numerics_e2e.txt 2>/dev/null
cat $O/numerics_e2e.txt
python bench.py --steps 60 --cpu-budget-s 5 2>$O/bench.err | tail -1 > $O/bench_fused_with_second_line.json
python -c "
import json
d=json.load(open('$O/bench_fused_with_second_line.json')); print('headline', round(d['value'],1), d['config']['windows_tiles_per_s']); sl=d['second_line']; print('second', sl.get('value'), sl.get('error'), sl.get('windows_tiles_per_s'), {k:(round(v['ms_per_step'],2), round(v['frac'],3)) for k,v in sl['roofline']['per_kernel'].items()} if 'roofline' in sl else None)
"
tail -3 $O/bench.err
