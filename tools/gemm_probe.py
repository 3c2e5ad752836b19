#!/usr/bin/env python
'''GEMM diagnostics: (a) the square product with the launch capped to 1 / 2 / 3 workgroups per CU
(issue efficiency at 1 / 2 / 3 waves per SIMD), (b) hipBLASLt (torch.matmul) on the same shapes.
Run on a GPU box:  python tools/gemm_probe.py'''
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g  # noqa: E402

g.load_package()
from danet_amd import ops  # noqa: E402


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    dev = 'cuda'
    shapes = [('square NN', 4096, 4096, 4096, 0, 0), ('proj NN', 4096, 2580, 600, 0, 0),
              ('dYc NT', 4096, 600, 2580, 0, 1), ('dX NT', 4096, 600, 1200, 0, 1),
              ('gx NN', 4096, 1200, 600, 0, 0), ('dWout TN', 600, 2580, 4096, 1, 0)]
    full = '--sweep' in sys.argv
    for name, M, N, K, ta, tb in shapes:
        A = torch.randn((K, M) if ta else (M, K), device=dev)
        B = torch.randn((N, K) if tb else (K, N), device=dev)
        C = torch.empty(M, N, device=dev)
        fl = 2.0 * M * N * K

        def run(**kw):
            return timeit(lambda: ops.gemm(A, B, C, M, N, K, A.shape[1], B.shape[1], N,
                                           transA=ta, transB=tb, **kw))
        res = []
        if full:
            for wgs in (256, 512, 768):
                res.append(('wgs%d' % wgs, run(max_workgroups=wgs)))
        res.append(('tiles', run()))
        res.append(('stream-K', run(streamk=True)))
        At, Bt = (A.t() if ta else A), (B.t() if tb else B)
        res.append(('hipBLASLt', timeit(lambda: torch.matmul(At, Bt, out=C))))
        print('%-10s M=%4d N=%4d K=%4d  ' % (name, M, N, K) +
              '  '.join('%s %6.1f us %5.1f TF' % (n, us, fl / us / 1e6) for n, us in res))


if __name__ == '__main__':
    main()
