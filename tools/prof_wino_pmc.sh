#!/bin/bash
# PMC view of the Winograd kernels (tools/bench_wino.py): MFMA busy cycles of wino_gemm, HBM bytes of wino_input
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_MFMA" "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_wino/pass$i -o p -- python3 $R/tools/bench_wino.py > /dev/null 2>> $R/gpurun_out/prof_stderr.log
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$R/gpurun_out/pmc_wino/pass*/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        if 'wino' not in r['Kernel_Name']: continue
        agg[(r['Kernel_Name'][24:45], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
    print(f.split('/')[-2])
    for k, d in agg.items():
        print('  ', k, {n: round(sum(v) / len(v)) for n, v in d.items()})
PY
