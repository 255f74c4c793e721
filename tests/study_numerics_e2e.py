"""Study (checker-side script: imports oracle/, which only tests/ may; moved here from tools/r4 in round 6).
Whole-network error of the HIP path against the oracle (torch-CPU fp32 restatement, bit-identical to the reference on the goldens) under
the 3x3-convolution route selected by the environment (default F(4x4); LANEMAP_WINO_F44=0: direct MFMA).
Prints, per raw output, max |err|, the tensor scale and err / scale, and the decode-level errors on the G15 weight set (absolute)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden'))
from lanemapping_amd import synth  # noqa: E402
from lanemapping_amd.boundary import build_net_from_config  # noqa: E402
from oracle import net_ref, decode_ref  # noqa: E402

dev = torch.device('cuda:0')
route = 'direct' if os.environ.get('LANEMAP_WINO_F44', '1') == '0' else ('F(4x4) fp16x2 split' if os.environ.get('LANEMAP_WINO_SPLIT', '0') != '0' else 'F(4x4)')
for gains, label in (({}, 'seeded weights (G10 set)'), ({'heads.offset2.2.weight': 0.02, 'heads.offset2.2.bias': 0.02}, 'G15 weight set')):
    net = build_net_from_config('Proj_polyline_fpn_vit_vertex_2', device='cpu')
    synth.fill_module_(net, 2021)
    synth.apply_gains_(net, gains)
    sd = {k: v.clone() for k, v in net.state_dict().items()}
    net = net.to(dev)
    for seed in (2021, 2024):
        x = torch.from_numpy(synth.bev_batch([seed], 1152))
        with torch.no_grad():
            raw = net.forward_raw({'proj': x.to(dev)})
            ref = net_ref.detector_forward(sd, x)
        parts = []
        for k in ('proposal_conf', 'ext2', 'cls2', 'offset2', 'orient', 'semantic_seg', 'endp_est'):
            a, b = raw[k].float().cpu(), ref[k]
            err, scale = float((a - b).abs().max()), float(b.abs().max())
            parts.append(f'{k} {err:.2e} / {scale:.1f} = {err / max(scale, 1.0):.1e}')
        net({'proj': x.to(dev)})
        c = net.heads._compact
        d = decode_ref.decode_column_proposals({k: v.numpy() for k, v in ref.items()})
        eo = float((c['cls_offset'].cpu() - d['cls_offset']).abs().max())
        ec = float((c['prop_conf'].cpu() - d['prop_conf']).abs().max())
        print(f'{route:7s} {label}, tile {seed}: ' + '; '.join(parts) + f'; decode: cls_offset {eo:.2e}, prop_conf {ec:.2e} (absolute)', flush=True)
