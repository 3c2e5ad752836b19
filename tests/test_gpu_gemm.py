'''
GPU tests (run with -m gpu): the exact-fp32 GEMM schedules (stream-K, K-concatenated, capped groups) and ops.lyr_linear.
Shared helpers: tests/gpu_helpers.py.
'''
import numpy as np
import pytest
import torch
from gpu_helpers import TOL, check_lstm_status, cu, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _lstm_status():
    yield
    check_lstm_status()


@pytest.mark.parametrize('M,K,N,bias', [(37, 19, 23, True), (256, 129, 512, True), (8, 5, 3, False)])
def test_linear_fn_forward_backward(M, K, N, bias):
    '''ops.lyr_linear (app/ops.py:66-89) on the last axis of a 3-D tensor: y, dx, dW, db'''
    from danet_amd import ops
    rng = np.random.RandomState(M + N)
    x = rng.randn(2, M, K); W = rng.randn(K, N) * 0.3; b = rng.randn(N); dy = rng.randn(2, M, N)
    xt = cu(x).requires_grad_(True); Wt = cu(W).requires_grad_(True)
    bt = cu(b).requires_grad_(True) if bias else None
    y = ops.lyr_linear(xt, Wt, bt)
    want = x @ W + (b if bias else 0.0)
    assert relerr(y.detach().cpu().numpy(), want) < TOL
    y.backward(cu(dy))
    assert relerr(xt.grad.cpu().numpy(), dy @ W.T) < TOL
    assert relerr(Wt.grad.cpu().numpy(), np.einsum('bmk,bmn->kn', x, dy)) < TOL
    if bias:
        assert relerr(bt.grad.cpu().numpy(), dy.sum((0, 1))) < TOL


@pytest.mark.parametrize('M,N,K1,K2,ta,tb', [(4096, 600, 1200, 1200, 0, 1), (4096, 1200, 2400, 2400, 0, 1),
                                             (130, 70, 32, 500, 0, 0), (64, 300, 16, 17, 1, 0),
                                             (257, 129, 704, 90, 1, 1), (700, 2600, 48, 33, 0, 0)])
def test_gemm_streamk_kcat(M, N, K1, K2, ta, tb):
    '''danet_gemm_f32_streamk_kcat: C = A1 B1 + A2 B2 (+bias)(+beta C) on the hybrid stream-K
    schedule (whole tiles data-parallel, the ragged last round cut along K; the two operand
    pairs are one concatenated contraction): correct, bit-reproducible'''
    from danet_amd import ops
    rng = np.random.RandomState(M + N + K1 + K2)
    cu = lambda a: torch.as_tensor(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    mk = lambda K: (rng.randn(K, M) if ta else rng.randn(M, K), rng.randn(N, K) if tb else rng.randn(K, N))
    (A1, B1), (A2, B2) = mk(K1), mk(K2)
    op = lambda A, Bm: (A.T if ta else A) @ (Bm.T if tb else Bm)
    ref = op(A1, B1) + op(A2, B2)
    bias, C0 = rng.randn(N), rng.randn(M, N)
    d = [cu(x) for x in (A1, B1, A2, B2)]
    outs = []
    for _ in range(2):
        C = torch.empty(M, N, device='cuda')
        ops.gemm_kcat(d[0], d[0].shape[1], d[1], d[1].shape[1], K1, d[2], d[2].shape[1], d[3],
                      d[3].shape[1], K2, C, M, N, N, transA=ta, transB=tb, streamk=True)
        outs.append(C)
    err = np.abs(outs[0].cpu().numpy() - ref).max() / np.abs(ref).max()
    assert err < 2e-5, err
    assert torch.equal(outs[0], outs[1])
    C = cu(C0)
    ops.gemm_kcat(d[0], d[0].shape[1], d[1], d[1].shape[1], K1, d[2], d[2].shape[1], d[3],
                  d[3].shape[1], K2, C, M, N, N, transA=ta, transB=tb, bias=cu(bias), beta=1.0,
                  streamk=True)
    assert np.abs(C.cpu().numpy() - (ref + bias + C0)).max() / np.abs(ref).max() < 2e-5


@pytest.mark.parametrize('M,N,K', [(4096, 2580, 600), (4096, 600, 2580), (4096, 5160, 1200), (2100, 3000, 50),
                                   (1030, 1300, 70)])
def test_gemm_hybrid_streamk_many_tiles(M, N, K):
    '''products with more tiles than workgroups: every workgroup owns whole tiles and only the
    ragged last round is cut along K -- correct, reproducible, identical under a capped grid
    of a different size only up to summation order'''
    from danet_amd import ops
    rng = np.random.RandomState(M + N + K)
    A = torch.as_tensor(rng.randn(M, K).astype(np.float32)).cuda()
    Bm = torch.as_tensor(rng.randn(K, N).astype(np.float32)).cuda()
    ref = (A.double() @ Bm.double()).cpu().numpy()
    outs = []
    for _ in range(2):
        C = torch.full((M, N), float('nan'), device='cuda')
        ops.gemm(A, Bm, C, M, N, K, K, N, N, streamk=True)
        outs.append(C)
    assert np.abs(outs[0].cpu().numpy() - ref).max() / np.abs(ref).max() < 2e-5
    assert torch.equal(outs[0], outs[1])
    C = torch.full((M, N), float('nan'), device='cuda')
    ops.gemm_group([(A, K, Bm, N, C, N, M, N, 0.0)], K, max_workgroups=104)
    assert np.abs(C.cpu().numpy() - ref).max() / np.abs(ref).max() < 2e-5


@pytest.mark.parametrize('K,shapes,ta,tb', [
    (4096, [(600, 1200), (300, 1200), (600, 1200), (300, 1200)], 1, 0),   # a layer's dW group
    (300, [(129, 70), (1, 5), (257, 300)], 0, 0), (90, [(64, 64), (130, 200)], 1, 1),
    (1000, [(500, 40)], 0, 1)])
def test_capped_group_on_the_short_mfma_instruction(K, shapes, ta, tb):
    '''capped stream-K groups (the ones that run beside a BPTT kernel) take the 16x16x4 k-loop
    (option gemm_mfma16): the same bits as the 32x32x2 loop (same k order per output element), and
    checked against float64'''
    from danet_amd import ops, _lib
    rng = np.random.RandomState(K + len(shapes))
    probs, refs, outs = [], [], []
    for i, (M, N) in enumerate(shapes):
        A = rng.randn(K, M) if ta else rng.randn(M, K)
        Bm = rng.randn(N, K) if tb else rng.randn(K, N)
        C0 = rng.randn(M, N)
        beta = float(i % 2)
        refs.append(((A.T if ta else A) @ (Bm.T if tb else Bm) + beta * C0, C0))
        dA, dB, dC = cu(A), cu(Bm), cu(C0)
        probs.append((dA, dA.shape[1], dB, dB.shape[1], dC, N, M, N, beta))
        outs.append(dC)
    res = {}
    for mf in (1, 0):
        _lib.set_option('gemm_mfma16', mf)
        for (ref, C0), dC in zip(refs, outs):
            dC.copy_(cu(C0))
        ops.gemm_group(probs, K, transA=ta, transB=tb, max_workgroups=256)
        torch.cuda.synchronize()
        res[mf] = [dC.clone() for dC in outs]
    for (ref, _), a, b in zip(refs, res[1], res[0]):
        assert relerr(a.cpu().numpy(), ref) < 2e-5
        assert torch.equal(a, b)       # both instructions are k-ordered fp32 fma chains: the same bits
