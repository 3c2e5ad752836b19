#!/usr/bin/env python
'''GEMM microbenchmark over the shapes of one cfg-2 train step (+ a square
reference shape).  Run on a GPU box:  python tools/bench_gemm.py'''
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.load_package()
from danet_amd import ops  # noqa: E402

SHAPES = [
    # name, M, N, K, transA, transB, count per step
    ('square', 4096, 4096, 4096, 0, 0, 0),
    ('gx l0   NN', 4096, 1200, 129, 0, 0, 2),
    ('gx l1-2 NN', 4096, 1200, 600, 0, 0, 4),
    ('proj    NN', 4096, 2580, 600, 0, 0, 1),
    ('dWout   TN', 600, 2580, 4096, 1, 0, 1),
    ('dYc     NT', 4096, 600, 2580, 0, 1, 1),
    ('dWx l12 TN', 600, 1200, 4096, 1, 0, 4),
    ('dWx l0  TN', 129, 1200, 4096, 1, 0, 2),
    ('dWh     TN', 300, 1200, 4096, 1, 0, 6),
    ('dX      NT', 4096, 600, 1200, 0, 1, 4),
]


def main():
    dev = 'cuda'
    tot_us, tot_fl = 0.0, 0.0
    for name, M, N, K, ta, tb, cnt in SHAPES:
        A = torch.randn((K, M) if ta else (M, K), device=dev)
        B = torch.randn((N, K) if tb else (K, N), device=dev)
        C = torch.empty(M, N, device=dev)
        for _ in range(3):
            ops.gemm(A, B, C, M, N, K, A.shape[1], B.shape[1], N, transA=ta, transB=tb)
        torch.cuda.synchronize()
        reps = 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            ops.gemm(A, B, C, M, N, K, A.shape[1], B.shape[1], N, transA=ta, transB=tb)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        fl = 2.0 * M * N * K
        print('%-12s M=%5d N=%5d K=%5d  %8.1f us  %6.1f TFLOP/s  x%d' % (
            name, M, N, K, us, fl / us / 1e6, cnt))
        tot_us += us * cnt
        tot_fl += fl * cnt
    print('per-step GEMM total: %.1f us, %.1f GFLOP, %.1f TFLOP/s' % (
        tot_us, tot_fl / 1e9, tot_fl / tot_us / 1e6))


if __name__ == '__main__':
    main()
