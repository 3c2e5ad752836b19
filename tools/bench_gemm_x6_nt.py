#!/usr/bin/env python
'''GPU box: the packed-weight products (csrc/gemm_x6.hip, NT) next to the exact-fp32 kernels on the
step's shapes: error against the float64 product and time.  python tools/bench_gemm_x6_nt.py'''
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.load_package()
from danet_amd import ops

SHAPES = [('proj', 4096, 2580, 600, 0), ('dYc', 4096, 600, 2580, 0), ('dX (kcat)', 4096, 600, 1200, 1200),
          ('proj cfg4', 4096, 5160, 600, 0), ('gx h600', 4096, 2400, 1200, 0), ('dX h600', 4096, 1200, 2400, 2400),
          ('square', 4096, 4096, 4096, 0), ('cfg5 gx', 1251, 1200, 600, 0), ('ragged', 257, 129, 20, 44)]


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
    C6, C32 = torch.empty(M, N, device='cuda'), torch.empty(M, N, device='cuda')
    x6 = lambda: ops.gemm_w(A1, K1, B1, K1, 1, C6, M, N, K1, N, A2=A2, lda2=K2, W2=B2, K2=K2, sn2=K2)
    if K2:
        f32 = lambda: ops.gemm_kcat(A1, K1, B1, K1, K1, A2, K2, B2, K2, K2, C32, M, N, N, transB=True, streamk=K1 % 16 == 0)
    else:
        f32 = lambda: ops.gemm(A1, B1, C32, M, N, K1, K1, K1, N, transB=True, streamk=(N <= 1200))
    x6(); f32()
    e6 = float((C6.double() - ref).abs().max() / ref.abs().max())
    e32 = float((C32.double() - ref).abs().max() / ref.abs().max())
    fl = 2.0 * M * N * (K1 + K2)
    t6, t32 = timeit(x6), timeit(f32)
    print('%-10s M=%5d N=%5d K=%5d+%-5d  x6 %7.1f us %6.1f TFLOP/s err %.1e | fp32 %7.1f us %6.1f TFLOP/s err %.1e | %.2fx'
          % (name, M, N, K1, K2, t6, fl / t6 / 1e6, e6, t32, fl / t32 / 1e6, e32, t32 / t6), flush=True)
