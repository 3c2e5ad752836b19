'''CPU study (numpy, no GPU): how accurate is an fp32 product emulated with split-bf16 operands?

The fp32 matrix cores of gfx950 peak at 157 TFLOP/s, the bf16 ones at 2.5 PFLOP/s (16 x).  Writing
every fp32 operand as hi + mid + lo with three bf16 pieces (8 significant bits each = 24) and
summing the n largest of the nine piece products in fp32 accumulators gives an "fp32" GEMM at
16 / n of the cost in MFMA time: n = 3 -> 5.3 x, n = 6 -> 2.7 x.  This script measures the error of
those emulations against float64 on the step's shapes (activations ~N(0,1) / tanh-like, weights
U(+-0.75/sqrt(300))), next to the error of a plain float32 product.  bf16 rounding is emulated by
round-to-nearest-even on the float32 bit pattern; piece products are exact in fp32 (8 x 8 bits), the
accumulation is float32 (numpy sgemm), like the MFMA's.

    python tools/bf16x_split_accuracy.py
'''
import numpy as np


def bf16(x):
    u = x.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.astype(np.uint32).view(np.float32)


def split3(x):
    hi = bf16(x)
    r = (x - hi).astype(np.float32)
    mid = bf16(r)
    lo = bf16((r - mid).astype(np.float32))
    return hi, mid, lo


def emulated(a, b, n_terms):
    A, B = split3(a), split3(b)
    order = [(0, 0), (0, 1), (1, 0), (0, 2), (2, 0), (1, 1), (1, 2), (2, 1), (2, 2)][:n_terms]
    out = np.zeros((a.shape[0], b.shape[1]), np.float32)
    for i, j in reversed(order):          # small terms first
        out += A[i] @ B[j]
    return out


def relerr(x, ref):
    return float(np.abs(x - ref).max() / np.abs(ref).max()), float(
        np.sqrt(((x - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean()))


def main():
    rng = np.random.RandomState(0)
    shapes = [('projection  [4096x600].[600x2580]', 4096, 600, 2580, 1.85),
              ('input half  [4096x600].[600x1200]', 4096, 600, 1200, 0.75 / np.sqrt(300)),
              ('weight grad [600x4096].[4096x1200]', 600, 4096, 1200, None)]
    print('%-38s %-12s %12s %12s' % ('shape', 'form', 'max rel', 'rms rel'))
    for name, M, K, N, wscale in shapes:
        a = np.tanh(rng.randn(M, K)).astype(np.float32)
        b = (rng.uniform(-wscale, wscale, (K, N)) if wscale else rng.randn(K, N) * 0.1).astype(np.float32)
        ref = a.astype(np.float64) @ b.astype(np.float64)
        rows = [('float32', a @ b)] + [('bf16 x %d' % n, emulated(a, b, n)) for n in (1, 3, 6, 9)]
        for form, out in rows:
            mx, rms = relerr(out.astype(np.float64), ref)
            print('%-38s %-12s %12.2e %12.2e' % (name, form, mx, rms))


if __name__ == '__main__':
    main()
