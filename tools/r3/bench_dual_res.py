import os, sys, torch
sys.path.insert(0, '/root/repo' if os.path.exists('/root/repo/lanemapping_amd') else os.getcwd())
from lanemapping_amd import ops
dev = torch.device('cuda:0'); B = 8
for cin, cout, dil, hw in [(64, 64, 1, 288)]:
    x = ops.new_act(B, cin, hw, hw, dev).normal_(); r = ops.new_act(B, cout, hw, hw, dev).normal_()
    w = torch.randn(cout, cin, 3, 3, device=dev) / (cin * 9) ** 0.5
    sc, sh = torch.rand(cout, device=dev) + 0.5, torch.randn(cout, device=dev)
    wu = ops.pack_wino(w); wf = ops.pack_wino_fragments(wu)
    y = ops.conv_wino_implicit(x, wf, cout, dil, scale=sc, shift=sh, res=r, act=ops.ACT_RELU)
    ok = torch.equal(y, ops.conv_wino(x, wu, cout, dil, scale=sc, shift=sh, res=r, act=ops.ACT_RELU))
    res = {}
    for name, kw in (('plain', {}), ('res', dict(scale=sc, shift=sh, res=r, act=ops.ACT_RELU))):
        for _ in range(2): ops.conv_wino_implicit(x, wf, cout, dil, out=y, **kw)
        torch.cuda.synchronize(); a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); a.record()
        for _ in range(20): ops.conv_wino_implicit(x, wf, cout, dil, out=y, **kw)
        b.record(); torch.cuda.synchronize(); res[name] = a.elapsed_time(b) / 20
    print(os.path.basename(os.environ.get('LANEMAP_HIP_LIB', 'product')), f'64->64@288 plain {res["plain"]:.3f} res {res["res"]:.3f}', 'ok' if ok else 'MISMATCH')
