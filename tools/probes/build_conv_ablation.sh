#!/bin/bash
# Ablation builds of conv_mfma.hip (NOT product): which phase of the slab loop costs MFMA issue slots?
#   A: no global prefetch in the loop        B: A + no LDS store      C: B + no LDS fragment reads (register-only MFMA)
set -e
cd "$(dirname "$0")/../.."
SRC=lanemapping_amd/csrc
OUT=tools/probes
OBJS=""
for f in errors.cpp conv_direct.hip norm_resize.hip vit.hip head.hip decode.hip raster.hip rowref.hip postproc.cpp; do OBJS="$OBJS lanemapping_amd/build/$f.o"; done
for v in A B C; do
  cp $SRC/conv_mfma.hip /tmp/conv_$v.hip
  sed -i 's|#include "common.h"|#include "'$PWD'/'$SRC'/common.h"|' /tmp/conv_$v.hip
  sed -i 's|        if (kt + 1 < KT) gload(kt + 1);   // next slab in flight under the MFMA block|        // ablated: no prefetch|' /tmp/conv_$v.hip
  if [ $v != A ]; then sed -i 's|        if (kt + 1 < KT) lstore(buf ^ 1);|        // ablated: no LDS store|' /tmp/conv_$v.hip; fi
  if [ $v = C ]; then
    sed -i 's|af\[i\] = \*reinterpret_cast<const f32x4\*>(Ab + i \* 32 \* LDS_LD + kk);|af[i] = f32x4{(float)kk, 1.f, 2.f, (float)lane};|; s|bf\[j\] = \*reinterpret_cast<const f32x4\*>(Bb + j \* 32 \* LDS_LD + kk);|bf[j] = f32x4{1.f, (float)kt, 3.f, (float)lane};|' /tmp/conv_$v.hip
  fi
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -x hip -c /tmp/conv_$v.hip -o /tmp/conv_$v.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT/liblanemap_ablate_$v.so /tmp/conv_$v.o $OBJS
done
ls -la $OUT/*.so
