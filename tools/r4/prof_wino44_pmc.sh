#!/bin/bash
# Matrix-core busy fraction of wino44_kernel from the PMC counters (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE), per layer shape of
# tools/r4/bench_wino44.py.  Counters only (no trace domains), as gpurun requires.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:?run through gpurun (GRAFT_REPO_ROOT is the repo copy on the GPU box)}
B=${1:-8}
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $R/gpurun_out/pmc_wino44 -o p -- python3 $R/tools/r4/bench_wino44.py $B 3 > /dev/null 2>> $R/gpurun_out/prof_stderr.log
python3 - <<PY
import csv, glob, collections
f = glob.glob('$R/gpurun_out/pmc_wino44/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    n = r['Kernel_Name']
    if 'wino44_kernel' not in n and 'wino_pipe' not in n and 'wino_dual' not in n: continue
    agg[(n.replace('(anonymous namespace)::', '').split('(')[0], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in sorted(agg.items()):
    m = {n: sum(v) / len(v) for n, v in d.items()}
    cyc = m['GRBM_GUI_ACTIVE'] / 8
    print(k, 'kernel cycles/XCD %.2fM' % (cyc / 1e6), 'waves %d' % m['SQ_WAVES'], 'mfma busy frac %.3f' % (m['SQ_VALU_MFMA_BUSY_CYCLES'] / 1024 / cyc),
          'VALU insts/wave %.0f' % (m.get('SQ_INSTS_VALU', 0) / max(m['SQ_WAVES'], 1)),
          'LDS bank-conflict frac %.3f' % (m.get('SQ_LDS_BANK_CONFLICT', 0) / max(m.get('SQ_LDS_IDX_ACTIVE', 1), 1)))
PY
