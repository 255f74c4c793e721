#!/bin/bash
# where do the wave cycles of wino_gemm / conv_mfma go?  (SQ wait / active breakdown, one PMC pass)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU --output-format csv -d $R/gpurun_out/pmc_wino2 -o p -- python3 $R/tools/bench_wino.py > /dev/null 2>> $R/gpurun_out/prof_stderr.log
python3 - <<PY
import csv, glob, collections
f = glob.glob('$R/gpurun_out/pmc_wino2/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'wino_gemm' not in n and 'conv_mfma' not in n: continue
    agg[(n[24:60], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    m = {n: sum(v) / len(v) for n, v in d.items()}
    w = m.get('SQ_WAVE_CYCLES', 1)
    print(k, ' '.join(f'{n[3:]}={v / w:.3f}' for n, v in m.items() if n != 'SQ_WAVE_CYCLES'))
PY
