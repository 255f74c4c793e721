#!/bin/bash
# PMC counters of the GroupNorm + ReLU + bilinear (+ sum, + 1x1) kernels (tools/bench_gn.py): where the waves' cycles go
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/pmc_gn
mkdir -p $O
i=0
for grp in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS" \
           "SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $O/pass$i -o p -- python3 $R/tools/bench_gn.py > /dev/null 2>> $R/gpurun_out/prof_stderr.log
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$O/pass*/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:70]
        if 'gn_relu' not in k: continue
        key = (k, r['Grid_Size'])
        agg[key][r['Counter_Name']] += float(r['Counter_Value'])
        n[key] += 1
    for key, d in agg.items():
        print(key, {c: '%.4g' % v for c, v in d.items()})
PY
