import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.load_package()
from danet_amd import ops, _lib
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
for name, M, N, K in (('proj', 4096, 2580, 600), ('proj cfg4', 4096, 5160, 600), ('gx h600', 4096, 2400, 1200)):
    gen = torch.Generator(device='cuda').manual_seed(1)
    A = torch.tanh(torch.randn(M, K, device='cuda', generator=gen))
    B = (torch.rand(N, K, device='cuda', generator=gen) - 0.5) * 0.1
    C = torch.empty(M, N, device='cuda')
    f = lambda: ops.gemm_w(A, K, B, K, 1, C, M, N, K, N)
    for rnd in range(4):
        row = []
        for plan in (0, 2 + 16, 3 + 16, 4 + 16):
            _lib.set_option('gemm_x6_plan', plan)
            row.append('%d:%6.1f' % (plan & 15, timeit(f)))
        print(name, '  '.join(row), flush=True)
_lib.set_option('gemm_x6_plan', 0)
