#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/gpurun_out/pmc_wino3 -o p -- python3 $R/tools/bench_wino.py > /dev/null 2>> $R/gpurun_out/prof_stderr.log
python3 - <<PY
import csv, glob, collections
f = glob.glob('$R/gpurun_out/pmc_wino3/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'wino_gemm' not in n and 'conv_mfma' not in n: continue
    agg[(n[24:60], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    m = {n: sum(v) / len(v) for n, v in d.items()}
    cyc = m['GRBM_GUI_ACTIVE'] / 8
    print(k, 'kernel cycles/XCD %.2fM' % (cyc / 1e6), 'avg resident waves %.0f' % (m['SQ_WAVE_CYCLES'] / cyc), 'waves launched %d' % m['SQ_WAVES'], 'mfma busy frac %.3f' % (m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc))
PY
