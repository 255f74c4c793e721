#!/bin/bash
# Device ISA of conv_wino44.hip + a per-barrier-interval count of scratch traffic, MFMAs and packed VALU (round-4 kernel work).
cd /root/repo/lanemapping_amd
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize "$@" -x hip --cuda-device-only -S csrc/conv_wino44.hip -o /tmp/w44.s -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A12 "wino44_kernelE" | grep -i "spill\|scratch\|VGPRs\|SGPRs"
awk '/^_ZN12_GLOBAL__N_113wino44_kernel/,/s_endpgm/' /tmp/w44.s > /tmp/w44k.s
awk '/s_barrier/{nb++; print "---- barrier", nb, "line", NR, " scratch ld/st:", ld+0, st+0, " mfma:", mf+0, " pk:", pk+0, " valu:", va+0, " ds:", ds+0; ld=0; st=0; mf=0; pk=0; va=0; ds=0} /scratch_load/{ld++} /scratch_store/{st++} /v_mfma/{mf++} /v_pk_/{pk++} /^\tv_/{va++} /^\tds_/{ds++} END{print "end", ld+0, st+0, mf+0, pk+0, va+0, ds+0}' /tmp/w44k.s
