#!/usr/bin/env python
'''GPU box: PROTOTYPE v3 (weight in operand layout, loaded past LDS) of fp32 products on the bf16 matrix cores (tools/csrc/gemm_x6b.hip, built on
demand with hipcc) next to the product's exact-fp32 kernels on the step's NT shapes: error against the
float64 product and time.  python tools/bench_gemm_x6b.py'''
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g
g.load_package()
from danet_amd import ops, _lib



import ctypes, subprocess
SRC = os.path.join(ROOT, 'tools', 'csrc', 'gemm_x6b.hip')
LIB = os.environ.get('X6B_LIB', os.path.join(ROOT, 'tools', 'csrc', 'libgemm_x6b.so'))
if 'X6B_LIB' not in os.environ and (not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC)):
    subprocess.check_call(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared',
                           '-I' + os.path.join(ROOT, 'include'), '-I' + os.path.join(ROOT, 'danet-tensorflow_amd', 'csrc'),
                           '-x', 'hip', SRC, '-o', LIB])
X = ctypes.CDLL(LIB)
c_p, c_i, c_l = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64
X.danet_pack3_bytes.restype = ctypes.c_size_t
X.danet_pack3_bytes.argtypes = [c_i, c_i]
X.danet_pack3_bf16.argtypes = [c_p, c_i, c_i, c_p, c_l, c_l, c_p]
X.danet_gemm_x6b_nt.argtypes = [c_p, c_i, c_i, c_i, c_p, c_i, c_p, c_i, c_p, c_i, c_p, c_p, c_i, c_p, ctypes.c_size_t]
_ws = torch.empty(64 << 20, dtype=torch.uint8, device='cuda')


def pack3(B, N, K, sn, sk):
    out = torch.empty(X.danet_pack3_bytes(N, K), dtype=torch.uint8, device=B.device)
    assert X.danet_pack3_bf16(_lib.stream(), N, K, B.data_ptr(), sn, sk, out.data_ptr()) == 0
    return out


def x6(A1, B1p, C, M, N, K1, A2=None, B2p=None, K2=0):
    p = lambda t: t.data_ptr() if t is not None else None
    rc = X.danet_gemm_x6b_nt(_lib.stream(), M, N, K1, A1.data_ptr(), K1, B1p.data_ptr(),
                             K2, p(A2), K2, p(B2p), C.data_ptr(), N, _ws.data_ptr(), _ws.numel())
    assert rc == 0, rc


SHAPES = [('proj  (W^T)', 4096, 2580, 600, 0), ('dYc', 4096, 600, 2580, 0), ('dX (kcat)', 4096, 600, 1200, 1200),
          ('proj cfg4', 4096, 5160, 600, 0), ('dX h600', 4096, 1200, 2400, 2400), ('square', 4096, 4096, 4096, 0),
          ('ragged', 257, 129, 20, 44)]


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
    B1p = pack3(B1, N, K1, K1, 1)
    B2p = pack3(B2, N, K2, K2, 1) if K2 else None
    x6(A1, B1p, C6, M, N, K1, A2, B2p, K2)
    if K2:
        f32 = lambda: ops.gemm_kcat(A1, K1, B1, K1, K1, A2, K2, B2, K2, K2, C32, M, N, N, transB=True, streamk=K1 % 16 == 0)
    else:
        f32 = lambda: ops.gemm(A1, B1, C32, M, N, K1, K1, K1, N, transB=True, streamk=(N <= 1200))
    f32()
    e6 = float((C6.double() - ref).abs().max() / ref.abs().max())
    e32 = float((C32.double() - ref).abs().max() / ref.abs().max())
    fl = 2.0 * M * N * (K1 + K2)
    t6 = timeit(lambda: x6(A1, B1p, C6, M, N, K1, A2, B2p, K2))
    tsplit = timeit(lambda: pack3(B1, N, K1, K1, 1))
    t32 = timeit(f32)
    print('%-12s M=%5d N=%5d K=%5d+%-5d  x6 %7.1f us %6.1f TFLOP/s err %.1e | fp32 %7.1f us %6.1f TFLOP/s err %.1e | %.2fx | pack(B1) %.1f us'
          % (name, M, N, K1, K2, t6, fl / t6 / 1e6, e6, t32, fl / t32 / 1e6, e32, t32 / t6, tsplit), flush=True)
