#!/bin/bash
# Experimental builds of conv_mfma.hip with -DLM_CONV_PIPE=<v> (NOT product): tools/probes/libconv_v<v>.so
set -e
cd "$(dirname "$0")/../.."
python -m lanemapping_amd.build > /dev/null 2>&1 || true
OBJS=""
for f in errors.cpp conv_wino.hip conv_direct.hip norm_resize.hip vit.hip head.hip decode.hip raster.hip rowref.hip lidar.hip postproc.cpp backproject.cpp; do OBJS="$OBJS lanemapping_amd/build/$f.o"; done
# usage: build_conv_variants.sh name:"-DFLAG=.. -DFLAG2=.." ...
for spec in "$@"; do
  v=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden $flags -x hip -c lanemapping_amd/csrc/conv_mfma.hip -o /tmp/conv_v$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/libconv_v$v.so /tmp/conv_v$v.o $OBJS
done
ls -la tools/probes/libconv_v*.so
