#!/bin/bash
# Where wino44_kernel's LDS bank conflicts are: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE per launch for the product and for the timing-ablation
# builds that compile one LDS user out (tools/build_variant.sh abl_not = no input transform: no raw-patch ds_read_b64, no V stores;
# abl_noepi = no epilogue exchange; abl_bare = MFMAs, A-fragment reads and barriers only).  Counters only (no trace domains), one pass.
R=${GRAFT_REPO_ROOT:?run through gpurun}
cd /tmp && export TMPDIR=/tmp
for v in ${W44_LDS_VARIANTS:-product abl_not abl_noepi abl_bare}; do
  if [ $v = product ]; then L=$R/lanemapping_amd/liblanemap_hip.so; else L=$R/tools/probes/lib_$v.so; fi
  rm -rf $R/gpurun_out/pmc_w44lds_$v
  LANEMAP_HIP_LIB=$L rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES GRBM_GUI_ACTIVE SQ_ACTIVE_INST_LDS --output-format csv \
      -d $R/gpurun_out/pmc_w44lds_$v -o p -- python3 $R/tools/r5/w44_time.py 16 > /dev/null 2>> $R/gpurun_out/prof_stderr.log
  python3 - <<PY
import csv, glob, collections
fs = glob.glob('$R/gpurun_out/pmc_w44lds_$v/**/*counter_collection.csv', recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(fs[0])):
    if 'wino44_kernel' not in r['Kernel_Name']: continue
    agg[r['Grid_Size']][r['Counter_Name']].append(float(r['Counter_Value']))
print('## $v')
for k, d in sorted(agg.items(), key=lambda kv: -int(kv[0])):
    m = {n: sum(x) / len(x) for n, x in d.items()}
    w = max(m['SQ_WAVES'], 1)
    print('grid %8s  waves %6d  LDS insts/wave %6.0f  LDS-array cycles/wave %7.0f  conflict cycles/wave %7.0f  (frac %.3f)  kernel cycles/XCD %.3fM' % (
        k, w, m.get('SQ_INSTS_LDS', 0) / w, m['SQ_LDS_IDX_ACTIVE'] / w, m['SQ_LDS_BANK_CONFLICT'] / w,
        m['SQ_LDS_BANK_CONFLICT'] / max(m['SQ_LDS_IDX_ACTIVE'], 1), m['GRBM_GUI_ACTIVE'] / 8 / 1e6))
PY
done
