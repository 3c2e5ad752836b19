#!/usr/bin/env python
'''GPU box: danet_gemm_x6 (csrc/gemm_x6.hip, NT) under every pinned plan -- rows per workgroup tile
(64 MI) x K slices, option gemm_x6_plan = MI + 16 * slices -- next to the modelled pick, on the
step's shapes: time and error against the float64 product.  python tools/bench_gemm_x6_plans.py'''
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.load_package()
from danet_amd import ops, _lib

SHAPES = [('proj', 4096, 2580, 600, 0), ('dYc', 4096, 600, 2580, 0), ('dX (kcat)', 4096, 600, 1200, 1200),
          ('proj cfg4', 4096, 5160, 600, 0), ('gx h600', 4096, 2400, 1200, 0), ('dX h600', 4096, 1200, 2400, 2400),
          ('dYc cfg4', 4096, 1200, 5160, 0), ('square', 4096, 4096, 4096, 0), ('cfg5 gx', 1251, 1200, 600, 0),
          ('ragged', 257, 129, 20, 44)]
if len(sys.argv) > 1:
    SHAPES = [s for s in SHAPES if any(a in s[0] for a in sys.argv[1:])]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, M, N, K1, K2 in SHAPES:
    gen = torch.Generator(device='cuda').manual_seed(M + N + K1)
    A1 = torch.tanh(torch.randn(M, K1, device='cuda', generator=gen))
    B1 = (torch.rand(N, K1, device='cuda', generator=gen) - 0.5) * 0.1
    A2 = B2 = None
    ref = A1.double() @ B1.double().t()
    if K2:
        A2, B2 = torch.randn(M, K2, device='cuda', generator=gen), torch.randn(N, K2, device='cuda', generator=gen)
        ref = ref + A2.double() @ B2.double().t()
    C6 = torch.empty(M, N, device='cuda')
    x6 = lambda: ops.gemm_w(A1, K1, B1, K1, 1, C6, M, N, K1, N, A2=A2, lda2=K2, W2=B2, K2=K2, sn2=K2)
    fl = 2.0 * M * N * (K1 + K2)
    nkt = (K1 + 15) // 16 + (K2 + 15) // 16
    row = []
    _lib.set_option('gemm_x6_plan', 0)
    C6.zero_(); x6()
    e0 = float((C6.double() - ref).abs().max() / ref.abs().max())
    t0 = timeit(x6)
    best = (1e9, None)
    for mi in (2, 3, 4):
        for s in (1, 2, 3, 4, 5, 6):
            if s > 1 and nkt // s < 8:
                continue
            _lib.set_option('gemm_x6_plan', mi + 16 * s)
            C6.zero_(); x6()
            e = float((C6.double() - ref).abs().max() / ref.abs().max())
            t = timeit(x6)
            row.append('%d/%d %6.1f%s' % (mi, s, t, '' if e < 4e-6 else ' ERR %.1e' % e))
            if t < best[0]:
                best = (t, (mi, s))
    _lib.set_option('gemm_x6_plan', 0)
    print('%-10s M=%5d N=%5d K=%5d+%-5d  modelled %7.1f us %6.1f TFLOP/s err %.1e | best %s %7.1f us %6.1f TFLOP/s'
          % (name, M, N, K1, K2, t0, fl / t0 / 1e6, e0, best[1], best[0], fl / best[0] / 1e6), flush=True)
    for i in range(0, len(row), 6):
        print('      ' + '  '.join(row[i:i + 6]), flush=True)
