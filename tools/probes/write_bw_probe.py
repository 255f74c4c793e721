import torch, sys, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from lanemapping_amd import ops
dev = torch.device('cuda:0')
x = torch.empty(2720 * 1024 * 1024 // 4, device=dev)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = timeit(lambda: x.fill_(1.0)); print(f'fill 2.85 GB: {ms:.3f} ms = {x.numel()*4/ms/1e9:.2f} TB/s')
ms = timeit(lambda: x.zero_()); print(f'zero (memset) 2.85 GB: {ms:.3f} ms = {x.numel()*4/ms/1e9:.2f} TB/s')
a = ops.new_act(8, 256, 288, 288, dev).normal_()
ms = timeit(lambda: ops.wino_transform(a, 1)); print(f'wino_input 256ch@288^2 B8: {ms:.3f} ms = {(2.72+0.68)/ms:.2f} TB/s (r+w)')
