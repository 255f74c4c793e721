#!/bin/bash
# Sanitizer build of the HOST C++ of liblanemap_hip.so (SURVEY.md §5 "race detection / sanitizers"; CPU only - GPU ASan is not available on
# this pool): errors.cpp + the six sources that parse untrusted files (PNG, LAS header, JSON text) or run greedy index-heavy loops
# (polyline assembly, cross-tile merge, skeleton thinning, back-projection) compiled with g++ -fsanitize=address,undefined and linked
# with the product's own (uninstrumented) HIP objects, so that the library still exports the whole C-ABI and the existing CPU tests can
# run against it unchanged (LANEMAP_HIP_LIB).  -> tests/asan/_build/libhost_asan.so
set -e
cd "$(dirname "$0")/../.."
python -m lanemapping_amd.build > /dev/null
B=tests/asan/_build
mkdir -p $B
HOST="errors postproc merge_lines png_reader lane_json skeleton backproject"
for f in $HOST; do
  flags=""
  case $f in backproject|merge_lines) flags="-ffp-contract=off";; esac      # (EXACT_FP sources of lanemapping_amd/build.py)
  g++ -std=c++17 -O1 -g -fPIC -fvisibility=hidden -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer $flags \
      -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -w -c lanemapping_amd/csrc/$f.cpp -o $B/$f.o
done
OBJS=$(python -c "
from lanemapping_amd.build import SOURCES
host = set('$HOST'.split())
print(' '.join('lanemapping_amd/build/%s.o' % s for s in SOURCES if s.rsplit('.', 1)[0] not in host))")
g++ -shared -fPIC -o $B/libhost_asan.so $(for f in $HOST; do echo $B/$f.o; done) $OBJS -L/opt/rocm/lib -lamdhip64 -fsanitize=address,undefined
echo $B/libhost_asan.so
