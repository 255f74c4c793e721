#!/bin/bash
# Timing-ablation builds of one source: tools/build_variant.sh NAME SOURCE "-DFLAG ..."  ->  tools/probes/lib_NAME.so
# (all other objects come from lanemapping_amd/build; run with LANEMAP_HIP_LIB=tools/probes/lib_NAME.so)
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; FLAGS=$3
EXTRA=""
[ "$SRC" = conv_wino44.hip ] && EXTRA="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden $EXTRA $FLAGS -x hip -c lanemapping_amd/csrc/$SRC -o tools/probes/${SRC}_$NAME.o
OBJS=$(ls lanemapping_amd/build/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/lib_$NAME.so $OBJS tools/probes/${SRC}_$NAME.o -lz
rm tools/probes/${SRC}_$NAME.o
echo built tools/probes/lib_$NAME.so
