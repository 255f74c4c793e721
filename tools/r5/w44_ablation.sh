#!/bin/bash
# Timing ablations of wino44_kernel (tools/build_variant.sh abl_<name> conv_wino44.hip -DLM_QABL_...): per-layer ms of every variant, REG=0 route
R=${GRAFT_REPO_ROOT:?run through gpurun}
cd $R
for v in "" abl_nomid abl_not abl_nob abl_noglds abl_bare abl_noepi abl_nores abl_nostore; do
  if [ -z "$v" ]; then L=$R/lanemapping_amd/liblanemap_hip.so; else L=$R/tools/probes/lib_$v.so; fi
  echo "## ${v:-product}"
  LANEMAP_HIP_LIB=$L python tools/r5/w44_time.py 16 2>/dev/null
done
