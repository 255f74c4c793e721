#!/bin/bash
# PMC passes over tools/bench_conv.py (one counter group per run; never combined with tracing). Output: gpurun_out/pmc_conv/passN
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $grp --output-format csv -d $R/gpurun_out/pmc_conv/pass$i -o p -- python3 $R/tools/bench_conv.py > /dev/null 2>> $R/gpurun_out/prof_stderr.log
done
python3 - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$R/gpurun_out/pmc_conv/pass*/*counter_collection.csv')):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        if 'conv_mfma' not in r['Kernel_Name']: continue
        key = (r['Grid_Size'], r['LDS_Block_Size'] if 'LDS_Block_Size' in r else '')
        agg[key][r['Counter_Name']] += float(r['Counter_Value']); cnt[(key, r['Counter_Name'])] += 1
    print(f.split('/')[-2])
    for key, d in agg.items():
        print('  grid', key, {k: round(v / cnt[(key, k)]) for k, v in d.items()})
PY
