#!/bin/bash
# Experimental builds (NOT product): compile ONE source with extra -D flags and link it with the product's other objects.
# usage: build_variants.sh <source.hip> name:"-DFLAG=.. -DFLAG2=.." ...   ->  tools/probes/libvar_<name>.so
set -e
cd "$(dirname "$0")/../.."
python -m lanemapping_amd.build > /dev/null 2>&1 || true
SRC=$1; shift
OBJS=""
for f in errors.cpp conv_mfma.hip conv_wino.hip conv_direct.hip norm_resize.hip vit.hip head.hip decode.hip raster.hip rowref.hip lidar.hip postproc.cpp backproject.cpp png_reader.cpp lane_json.cpp; do
  [ "$f" = "$SRC" ] || OBJS="$OBJS lanemapping_amd/build/$f.o"
done
for spec in "$@"; do
  v=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden $flags -x hip -c lanemapping_amd/csrc/$SRC -o /tmp/var_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/libvar_$v.so /tmp/var_$v.o $OBJS -lz
done
ls -la tools/probes/libvar_*.so
