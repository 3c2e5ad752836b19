#!/usr/bin/env python
'''GPU box: the weight-gradient groups (TN, both operands activations) on the bf16 matrix cores
(csrc/gemm_x6.hip) next to the exact-fp32 stream-K group: time alone on the GPU.
python tools/bench_gemm_x6_tn.py'''
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.load_package()
from danet_amd import ops


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


GROUPS = [('layer (D 600, H 300) x 2 dirs', 4096, [(600, 1200), (600, 1200), (300, 1200), (300, 1200)]),
          ('bottom layer (D 129)', 4096, [(129, 1200), (129, 1200), (300, 1200), (300, 1200)]),
          ('dWout', 4096, [(600, 2580)]),
          ('cfg4 H 600 layer', 4096, [(1200, 2400), (1200, 2400), (600, 2400), (600, 2400)])]
for name, K, shapes in GROUPS:
    probs = []
    for M, N in shapes:
        lda = (M + 3) // 4 * 4
        probs.append((torch.randn(K, lda, device='cuda'), lda, torch.randn(K, N, device='cuda'), N,
                      torch.zeros(M, N, device='cuda'), N, M, N, 0.0))
    fl = sum(2.0 * M * N * K for M, N in shapes)
    res = {}
    for mode in (3, 1):
        ops.GEMM_X6 = mode
        for wgs in (512, 256):
            res[(mode, wgs)] = timeit(lambda: ops.gemm_group(probs, K, transA=True, max_workgroups=wgs))
    ref = [(p[0][:, :p[6]].double().t() @ p[2].double()) for p in probs]
    ops.GEMM_X6 = 3
    ops.gemm_group(probs, K, transA=True)
    e6 = max(float((p[4].double() - r).abs().max() / r.abs().max()) for p, r in zip(probs, ref))
    ops.GEMM_X6 = 1
    ops.gemm_group(probs, K, transA=True, max_workgroups=512)
    e32 = max(float((p[4].double() - r).abs().max() / r.abs().max()) for p, r in zip(probs, ref))
    print('%-32s %5.1f GFLOP  x6 %6.1f us %6.1f TFLOP/s err %.1e | fp32 (512 wgs) %6.1f us %6.1f TFLOP/s err %.1e, (256 wgs) %6.1f us | %.2fx'
          % (name, fl / 1e9, res[(3, 512)], fl / res[(3, 512)] / 1e6, e6, res[(1, 512)], fl / res[(1, 512)] / 1e6, e32,
             res[(1, 256)], res[(1, 512)] / res[(3, 512)]), flush=True)
