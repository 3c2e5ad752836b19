#!/usr/bin/env python
'''GPU box: danet_gemm_x6 (csrc/gemm_x6.hip, NT) with and without the HYBRID schedule of round 6 -- the
whole rounds of tiles computed whole, only the ragged remainder cut along K -- on the step's shapes with
more than 512 tiles: time and error against the float64 product under option gemm_x6_plan =
131072 (never hybrid) / 0 (modelled) / 65536 + 2 + 16 s (hybrid pinned at s slices).
python tools/bench_gemm_x6_hybrid.py'''
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.load_package()
from danet_amd import ops, _lib

SHAPES = [('proj', 4096, 2580, 600, 0), ('proj cfg4', 4096, 5160, 600, 0), ('gx h600', 4096, 2400, 1200, 0),
          ('dX h600', 4096, 1200, 2400, 2400), ('dYc', 4096, 600, 2580, 0), ('square', 4096, 4096, 4096, 0)]


def timeit(fn, reps=30):
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
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    row = []
    for label, pin in (('plain', 131072), ('auto', 0), ('hyb2', 65536 + 2 + 32), ('hyb3', 65536 + 2 + 48),
                       ('hyb4', 65536 + 2 + 64)):
        _lib.set_option('gemm_x6_plan', pin)
        C6.zero_(); x6()
        e = float((C6.double() - ref).abs().max() / ref.abs().max())
        first = C6.clone()
        t = timeit(x6)
        assert torch.equal(first, C6)          # bit-reproducible run to run
        row.append('%s %6.1f us %5.1f TF err %.1e' % (label, t, fl / t / 1e6, e))
    _lib.set_option('gemm_x6_plan', 0)
    print('%-10s %4d tiles | %s' % (name, tiles, ' | '.join(row)), flush=True)
