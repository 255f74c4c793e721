"""In-kernel phase profile of wino44_kernel (needs the -DLM_QPROF probe build: tools/build_variant.sh qprof conv_wino44.hip -DLM_QPROF,
run with LANEMAP_HIP_LIB=tools/probes/lib_qprof.so).  Prints shader-clock cycles per workgroup (wave 0) and phase."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', '..'))
from lanemapping_amd import ops  # noqa: E402

NAMES = ['set-up', 'prologue', 'transform', 'barrier after T', 'MFMA phases', 'barrier after M', 'patch-load wait', 'drain', 'epi set-up',
         'epi residual issue', 'epi barrier 1', 'TOTAL', 'epi product stores', 'epi barrier 2', 'epi transform + tail', '-']
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
dev = torch.device('cuda:0')
lib = ops.lib()
lib.lm_qprof_read.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
for cin, cout, hw, dil in [(256, 256, 288, 1), (256, 256, 144, 2), (128, 128, 144, 1), (64, 64, 288, 1)]:
    x = ops.new_act(B, cin, hw, hw, dev).normal_()
    w = torch.randn((cout, cin, 3, 3), device=dev) / (cin * 9) ** 0.5
    wf = ops.pack_wino44_fragments_split(ops.pack_wino44(w)) if os.environ.get('QPROF_SPLIT') else ops.pack_wino44_fragments(ops.pack_wino44(w))
    res = None if os.environ.get('QPROF_NORES') else ops.new_act(B, cout, hw, hw, dev).normal_()
    y = ops.new_act(B, cout, hw, hw, dev)
    ops.conv_wino44(x, wf, cout, dil, res=res, act=ops.ACT_RELU, out=y)
    torch.cuda.synchronize()
    buf = (C.c_ulonglong * 17)()
    lib.lm_qprof_read(buf, 1)
    ops.conv_wino44(x, wf, cout, dil, res=res, act=ops.ACT_RELU, out=y)
    torch.cuda.synchronize()
    lib.lm_qprof_read(buf, 1)
    n = max(1, buf[16])
    print(f'{cin}->{cout} d{dil}@{hw} B{B}: {n} workgroup records; cycles per workgroup: ' +
          ', '.join(f'{NAMES[k]} {buf[k] / n:.0f}' for k in range(15)), flush=True)
    if hasattr(lib, 'lm_qgap_report'):      # round 5: idle time of a CU between two workgroups of that launch
        g = (C.c_double * 8)()
        lib.lm_qgap_report.argtypes = [C.POINTER(C.c_double)]
        lib.lm_qgap_report(g)
        print(f'    {g[0]:.0f} workgroups on {g[1]:.0f} CUs: {g[2]:.0f} cycles per workgroup; gap between two workgroups of a CU mean {g[3]:.0f}, median {g[4]:.0f}, '
              f'p90 {g[5]:.0f} cycles; launch span {g[6]:.0f} cycles, CUs busy {g[7]:.3f} of it', flush=True)
