#!/bin/bash
# Experimental builds (NOT product): compile ONE source with extra -D flags and link it with the product's other objects.
# usage: build_variants.sh <source.hip> name:"-DFLAG=.. -DFLAG2=.." ...   ->  tools/probes/libvar_<name>.so   (select with LANEMAP_HIP_LIB)
set -e
cd "$(dirname "$0")/../.."
python -m lanemapping_amd.build > /dev/null 2>&1 || true
SRC=$1; shift
OBJS=""
for f in $(python -c "from lanemapping_amd.build import SOURCES; print(' '.join(SOURCES))"); do
  [ "$f" = "$SRC" ] || OBJS="$OBJS lanemapping_amd/build/$f.o"
done
EXTRA=$(python -c "from lanemapping_amd.build import EXTRA_FLAGS, EXACT_FP; import sys; s=sys.argv[1]; print(' '.join(EXTRA_FLAGS.get(s, []) + (['-ffp-contract=off'] if s in EXACT_FP else [])))" "$SRC")
for spec in "$@"; do
  v=${spec%%:*}; flags=${spec#*:}
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden $EXTRA $flags -x hip -c lanemapping_amd/csrc/$SRC -o /tmp/var_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/probes/libvar_$v.so /tmp/var_$v.o $OBJS -lz
done
ls -la tools/probes/libvar_*.so
