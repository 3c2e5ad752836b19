'''
GPU parity tests (run with `-m gpu` on an MI355X): every entry point of
libdanet_hip.so, through the Python boundary, against the CPU oracle
(oracle/danet_oracle.py, oracle/torch_ref.py) on identical seeded inputs.

Tolerance: 1e-4 relative to the tensor's max magnitude (north_star: "within
1e-4 relative fp32"; relative-to-max because saturated masks / exact zeros make
per-element relative error meaningless); integer outputs (frame counts,
argmin / permutation indices) must be exact.
'''
import os
import itertools

import numpy as np
import pytest
import torch

from oracle import danet_oracle as O
from oracle import torch_ref as R

pytestmark = pytest.mark.gpu

TOL = 1e-4
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'frontend_ref.npz')


def relerr(a, b):
    a = np.asarray(a, dtype=np.complex128 if np.iscomplexobj(a) or np.iscomplexobj(b) else np.float64)
    b = np.asarray(b, dtype=a.dtype)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def cu(x, dtype=torch.float32):
    return torch.as_tensor(np.asarray(x)).to('cuda', dtype)


@pytest.fixture(autouse=True)
def _lstm_status():
    yield
    from danet_amd import ops
    torch.cuda.synchronize()
    assert ops.lstm_status_ok(), 'persistent LSTM kernel reported a hand-off timeout'


def test_single_hip_runtime_loaded():
    '''libdanet_hip.so must share torch's libamdhip64 (one HIP runtime)'''
    from danet_amd import _lib
    _lib.load()
    torch.zeros(1, device='cuda')
    libs = set()
    with open('/proc/self/maps') as f:
        for line in f:
            if 'libamdhip64' in line:
                libs.add(line.split()[-1])
    assert len(libs) == 1, libs
    assert any('libdanet_hip.so' in l for l in open('/proc/self/maps'))


# ------------------------------------------------------------------ GEMM
@pytest.mark.parametrize('M,N,K', [(1, 1, 1), (7, 5, 3), (128, 128, 16), (130, 260, 33),
                                   (257, 129, 129), (64, 1200, 132), (300, 1200, 4096),
                                   (600, 2580, 1024), (4096, 600, 64)])
@pytest.mark.parametrize('ta,tb', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm(M, N, K, ta, tb):
    from danet_amd import ops
    rng = np.random.RandomState(M * 31 + N * 7 + K + ta * 2 + tb)
    A = rng.randn(K, M) if ta else rng.randn(M, K)
    Bm = rng.randn(N, K) if tb else rng.randn(K, N)
    bias = rng.randn(N)
    C0 = rng.randn(M, N)
    ref = (A.T if ta else A) @ (Bm.T if tb else Bm)
    dA, dB = cu(A), cu(Bm)
    C = torch.empty(M, N, device='cuda')
    ops.gemm(dA, dB, C, M, N, K, dA.shape[1], dB.shape[1], N, transA=ta, transB=tb)
    assert relerr(C.cpu().numpy(), ref) < 2e-5
    C = cu(C0)
    ops.gemm(dA, dB, C, M, N, K, dA.shape[1], dB.shape[1], N, transA=ta, transB=tb,
             bias=cu(bias), beta=1.0)
    assert relerr(C.cpu().numpy(), ref + bias + C0) < 2e-5


@pytest.mark.parametrize('M,N,K,ta,tb', [
    (1, 1, 1, 0, 0), (130, 260, 33, 1, 0), (4096, 600, 1200, 0, 1), (600, 1200, 4096, 1, 0),
    (129, 1200, 4096, 1, 0), (4096, 1200, 129, 0, 0), (300, 1200, 2000, 1, 1), (2100, 3000, 50, 0, 0)])
def test_gemm_streamk_schedule(M, N, K, ta, tb, monkeypatch):
    '''the opt-in stream-K schedule (danet_gemm_f32_streamk): correct, bit-reproducible
    across launches, under a concurrent load on another stream'''
    from danet_amd import ops
    monkeypatch.setattr(ops, 'STREAMK', 1)
    rng = np.random.RandomState(M + N + K)
    A = rng.randn(K, M) if ta else rng.randn(M, K)
    Bm = rng.randn(N, K) if tb else rng.randn(K, N)
    bias, C0 = rng.randn(N), rng.randn(M, N)
    ref = (A.T if ta else A) @ (Bm.T if tb else Bm)
    dA, dB = cu(A), cu(Bm)
    side = torch.cuda.Stream()
    noise = torch.randn(1024, 1024, device='cuda')
    outs = []
    for it in range(3):
        with torch.cuda.stream(side):
            for _ in range(4 * it):
                noise @ noise
        C = torch.empty(M, N, device='cuda')
        ops.gemm(dA, dB, C, M, N, K, dA.shape[1], dB.shape[1], N, transA=ta, transB=tb, streamk=True)
        outs.append(C)
    torch.cuda.synchronize()
    assert relerr(outs[0].cpu().numpy(), ref) < 2e-5
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    C = cu(C0)
    ops.gemm(dA, dB, C, M, N, K, dA.shape[1], dB.shape[1], N, transA=ta, transB=tb,
             bias=cu(bias), beta=1.0, streamk=True)
    assert relerr(C.cpu().numpy(), ref + bias + C0) < 2e-5


@pytest.mark.parametrize('K,shapes,ta,tb', [
    (4096, [(600, 1200), (300, 1200), (600, 1200), (300, 1200)], 1, 0),   # a layer's dW group
    (257, [(1, 1), (130, 260), (128, 128)], 0, 0),
    (64, [(300, 70), (5, 900), (129, 129), (64, 64), (1, 7), (256, 384)], 1, 1),
    (1000, [(2000, 40)], 0, 1)])
def test_gemm_streamk_grouped(K, shapes, ta, tb):
    '''danet_gemm_f32_streamk_grouped: several products in one launch, strided
    outputs, beta in {0,1}, bit-reproducible, capped grid'''
    from danet_amd import ops
    rng = np.random.RandomState(K + len(shapes))
    probs, refs, outs = [], [], []
    for i, (M, N) in enumerate(shapes):
        A = rng.randn(K, M) if ta else rng.randn(M, K)
        Bm = rng.randn(N, K) if tb else rng.randn(K, N)
        C0 = rng.randn(M, N + 3)                       # ldc > N
        beta = float(i % 2)
        ref = (A.T if ta else A) @ (Bm.T if tb else Bm) + beta * C0[:, :N]
        dA, dB, dC = cu(A), cu(Bm), cu(C0)
        probs.append((dA, dA.shape[1], dB, dB.shape[1], dC, N + 3, M, N, beta))
        refs.append((ref, C0)); outs.append(dC)
    for cap in (0, 64):
        res = []
        for rep in range(2):
            for (ref, C0), dC in zip(refs, outs):
                dC.copy_(cu(C0))
            ops.gemm_group(probs, K, transA=ta, transB=tb, max_workgroups=cap)
            res.append([dC.clone() for dC in outs])
        for (ref, C0), a, b in zip(refs, res[0], res[1]):
            N = ref.shape[1]
            assert relerr(a[:, :N].cpu().numpy(), ref) < 2e-5
            assert np.array_equal(a[:, N:].cpu().numpy(), C0[:, N:].astype(np.float32))   # padding untouched
            assert torch.equal(a, b)


@pytest.mark.parametrize('M,N,K1,K2,ta,tb', [(4096, 600, 1200, 1200, 0, 1), (130, 70, 33, 500, 0, 0),
                                             (64, 300, 16, 17, 1, 0), (257, 129, 700, 90, 1, 1)])
def test_gemm_kcat(M, N, K1, K2, ta, tb):
    '''ops.gemm_kcat without stream-K (two accumulating tile-kernel products): C = A1 B1 + A2 B2
    (+bias)(+beta C); the stream-K one-launch form is tested in test_gpu_gemm.py'''
    from danet_amd import ops
    rng = np.random.RandomState(M + N + K1 + K2)
    mk = lambda K: (rng.randn(K, M) if ta else rng.randn(M, K), rng.randn(N, K) if tb else rng.randn(K, N))
    (A1, B1), (A2, B2) = mk(K1), mk(K2)
    op = lambda A, Bm: (A.T if ta else A) @ (Bm.T if tb else Bm)
    ref = op(A1, B1) + op(A2, B2)
    bias, C0 = rng.randn(N), rng.randn(M, N)
    d = [cu(x) for x in (A1, B1, A2, B2)]
    C = torch.empty(M, N, device='cuda')
    ops.gemm_kcat(d[0], d[0].shape[1], d[1], d[1].shape[1], K1, d[2], d[2].shape[1], d[3], d[3].shape[1],
                  K2, C, M, N, N, transA=ta, transB=tb)
    assert relerr(C.cpu().numpy(), ref) < 2e-5
    C2 = cu(C0)
    ops.gemm_kcat(d[0], d[0].shape[1], d[1], d[1].shape[1], K1, d[2], d[2].shape[1], d[3], d[3].shape[1],
                  K2, C2, M, N, N, transA=ta, transB=tb, bias=cu(bias), beta=1.0)
    assert relerr(C2.cpu().numpy(), ref + bias + C0) < 2e-5


def test_gemm_strided_views_and_asymmetric():
    '''sub-matrix operands with ld > width; asymmetric operands catch a
    transposed fragment/epilogue mapping'''
    from danet_amd import ops
    M, N, K = 96, 72, 40
    A = np.zeros((M, K)); A[np.arange(min(M, K)), np.arange(min(M, K))] = 1.0
    Bm = np.arange(K * N, dtype=np.float64).reshape(K, N) / 100.0
    big_a = torch.zeros(M, K + 12, device='cuda'); big_a[:, 4:4 + K] = cu(A)
    big_b = torch.zeros(K, N + 8, device='cuda'); big_b[:, 8:8 + N] = cu(Bm)
    big_c = torch.full((M, N + 4), -7.0, device='cuda')
    ops.gemm(big_a[:, 4:], big_b[:, 8:], big_c[:, 4:], M, N, K, K + 12, N + 8, N + 4)
    got = big_c.cpu().numpy()
    assert relerr(got[:, 4:], A @ Bm) < 1e-6
    assert np.all(got[:, :4] == -7.0)


@pytest.mark.parametrize('M,N,K,ta,tb', [(130, 70, 129, 0, 0), (130, 70, 129, 0, 1), (129, 70, 130, 1, 0),
                                         (257, 131, 67, 1, 1), (64, 600, 2577, 0, 1)])
def test_gemm_ragged_operands_next_to_poison(M, N, K, ta, tb):
    '''The k-loop fetches 16-byte vectors at 4-byte alignment and relies on (a) the buffer range
    for what lies past an operand's LAST element and (b) masks for what lies past the end of a
    ROW: operands are packed into one buffer at odd (4-byte aligned only) offsets with NaN
    between and after them -- any element read from outside an operand poisons the product.'''
    from danet_amd import ops
    rng = np.random.RandomState(M + 3 * N + 5 * K + ta + 2 * tb)
    A = rng.randn(K, M) if ta else rng.randn(M, K)
    Bm = rng.randn(N, K) if tb else rng.randn(K, N)
    ref = (A.T if ta else A) @ (Bm.T if tb else Bm)
    pool = torch.full((A.size + Bm.size + 64,), float('nan'), device='cuda')
    oa = 5                                    # 20-byte offset: not 16-byte aligned
    ob = oa + A.size + 3
    pool[oa:oa + A.size] = cu(A).reshape(-1)
    pool[ob:ob + Bm.size] = cu(Bm).reshape(-1)
    dA = pool[oa:oa + A.size].view(*A.shape)
    dB = pool[ob:ob + Bm.size].view(*Bm.shape)
    C = torch.empty(M, N, device='cuda')
    ops.gemm(dA, dB, C, M, N, K, A.shape[1], Bm.shape[1], N, transA=ta, transB=tb)
    got = C.cpu().numpy()
    assert np.isfinite(got).all()
    assert relerr(got, ref) < 2e-5
    C2 = torch.empty(M, N, device='cuda')
    ops.gemm(dA, dB, C2, M, N, K, A.shape[1], Bm.shape[1], N, transA=ta, transB=tb, streamk=True)
    assert relerr(C2.cpu().numpy(), ref) < 2e-5
    # an operand whose LAST row ends exactly where its allocation ends
    tail_a = torch.empty(A.size, device='cuda'); tail_a.copy_(cu(A).reshape(-1))
    C3 = torch.empty(M, N, device='cuda')
    ops.gemm(tail_a.view(*A.shape), dB, C3, M, N, K, A.shape[1], Bm.shape[1], N, transA=ta, transB=tb)
    assert relerr(C3.cpu().numpy(), ref) < 2e-5


@pytest.mark.parametrize('M,N,K,ta,tb', [(4096, 600, 1200, 0, 1), (600, 1200, 4096, 1, 0),
                                         (300, 260, 600, 0, 0), (129, 1200, 4096, 1, 0), (200, 136, 1032, 1, 1)])
def test_gemm_dma_staging_is_bit_identical(M, N, K, ta, tb, monkeypatch):
    '''buffer_load ... lds staging (gemm_kloop_dma: swizzled / rotated LDS images, no staging
    registers) keeps the fragment order of the register path, so every launch flavour -- tiles,
    split-K, stream-K, the capped grouped launch -- returns the same bits with DANET_GEMM_DMA
    0 and 7'''
    from danet_amd import ops, _lib
    rng = np.random.RandomState(M + N + K)
    A = cu(rng.randn(K, M) if ta else rng.randn(M, K))
    Bm = cu(rng.randn(N, K) if tb else rng.randn(K, N))
    ref = ((A.T if ta else A).double() @ (Bm.T if tb else Bm).double()).cpu().numpy()
    outs = {}
    for mode in ('0', '7'):
        _lib.set_option('gemm_dma', int(mode))
        got = []
        C = torch.empty(M, N, device='cuda')
        ops.gemm(A, Bm, C, M, N, K, A.shape[1], Bm.shape[1], N, transA=ta, transB=tb)
        got.append(C.clone())
        ops.gemm(A, Bm, C, M, N, K, A.shape[1], Bm.shape[1], N, transA=ta, transB=tb, streamk=True)
        got.append(C.clone())
        ops.gemm_group([(A, A.shape[1], Bm, Bm.shape[1], C, N, M, N, 0.0)], K, transA=ta, transB=tb,
                       max_workgroups=256)
        got.append(C.clone())
        outs[mode] = got
        for g_ in got:
            assert relerr(g_.cpu().numpy(), ref) < 2e-5
    for a, b in zip(outs['0'], outs['7']):
        assert torch.equal(a, b)


def test_gemm_rejects_operands_of_2gib():
    '''operands are addressed through 32-bit buffer views: a span of 2 GiB or more is an
    argument error, not a wrong product (nothing is launched, the pointers are never read)'''
    from danet_amd import _lib
    L = _lib.load()
    t = torch.zeros(16, device='cuda')
    rc = L.danet_gemm_f32(_lib.stream(), 0, 0, 128, 128, 128, _lib.ptr(t), 1 << 23,
                          _lib.ptr(t), 128, _lib.ptr(t), 128, None, 0.0, None, 0, 0)
    assert rc == -1 and b'2 GiB' in L.danet_last_error()


# ------------------------------------------------------------------ centre
@pytest.mark.parametrize('B,T,D', [(2, 3, 5), (4, 16, 129), (32, 8, 600), (3, 640, 1200), (5, 130, 132)])
def test_center_layouts(B, T, D):
    from danet_amd import ops
    rng = np.random.RandomState(0)
    x = rng.randn(B, T, D) + 3.0
    ref = x - x.mean(axis=(1, 2), keepdims=True)
    ldo = (D + 3) // 4 * 4
    out = torch.full((T, B, ldo), 9.0, device='cuda')
    mean = ops.center(cu(x), B, T, D, 0, D, out, 1, ldo)
    got = out.cpu().numpy()
    assert relerr(got[:, :, :D].transpose(1, 0, 2), ref) < 1e-5
    assert np.all(got[:, :, D:] == 0.0)
    assert relerr(mean.cpu().numpy(), x.mean(axis=(1, 2))) < 1e-5
    back = torch.empty(B, T, D, device='cuda')
    ops.center(out, B, T, D, 1, ldo, back, 0, D)
    assert relerr(back.cpu().numpy(), ref) < 1e-5


def test_center_one_launch_equals_two_launches():
    '''B * 32 workgroups fit the GPU four to a CU -> both centring phases run in ONE launch
    (csrc/pointwise.hip center_fused_kernel); larger batches take the sum + apply launches.  Same
    partial sums in the same order: the same utterances give the same bits either way, call after
    call on the same scratch (the launch tags), with the scratch holding arbitrary bits.'''
    from danet_amd import ops, _lib
    L = _lib.load()
    rng = np.random.RandomState(5)
    T, D = 128, 600
    x = (rng.randn(40, T, D) * 2 + 1.5).astype(np.float32)
    xg = cu(x)
    big = torch.empty(T, 40, D, device='cuda')
    m40 = ops.center(xg, 40, T, D, 0, D, big, 1, D).clone()          # 1280 workgroups: two launches
    n = _lib.ws_bytes(_lib.WS_CENTER_MEAN, 32) // 4
    scratch = torch.full((n,), float('nan'), device='cuda')           # nothing in it needs initialising
    x32 = xg[:32].contiguous()
    for it in range(3):
        out = torch.full((T, 32, D), 7.0, device='cuda')
        rc = L.danet_center(_lib.stream(), 32, T, D, _lib.ptr(x32), 0, D, _lib.ptr(out), 1, D, _lib.ptr(scratch))
        assert rc == 0
        torch.cuda.synchronize()
        assert torch.equal(out, big[:, :32]), it
        assert torch.equal(scratch[:32], m40[:32]), it
    ref = x[:32].astype(np.float64)
    ref = ref - ref.mean(axis=(1, 2), keepdims=True)
    assert relerr(out.cpu().numpy().transpose(1, 0, 2), ref) < 1e-6


@pytest.mark.parametrize('B,train', [(32, True), (32, False), (8, True), (40, True)])
def test_encoder_prologue_equals_center_plus_prefill(B, train):
    '''danet_encoder_prologue = danet_center of the input + the recurrent launches' prefill riding in the
    same launch (B = 40: the two-launch centring form + a fill launch of its own): every buffer bit-equal to
    the separate calls'''
    from danet_amd import ops, _lib
    rng = np.random.RandomState(11)
    T, F, H, ndir, L = 128, 129, 300, 2, 3
    Fp = (F + 3) // 4 * 4
    x = cu((rng.randn(B, T, F) * 2 + 0.7).astype(np.float32))
    if train and _lib.load().danet_lstm_bwd_db_supported(T, B, H, ndir) != 1:
        pytest.skip('BPTT geometry outside the reduce-scatter kernel')

    def buffers():
        yp = [torch.zeros(T + 2, B, ndir * H, device='cuda') for _ in range(L)]
        n = _lib.ws_bytes(_lib.WS_LSTM, T, B, H, ndir)
        fw = [torch.zeros(n, dtype=torch.uint8, device='cuda') for _ in range(L)]
        bw = [torch.zeros(n, dtype=torch.uint8, device='cuda') for _ in range(L)] if train else None
        return yp, fw, bw
    yp1, fw1, bw1 = buffers()
    xc1 = torch.full((T, B, Fp), 5.0, device='cuda')
    ops.center(x, B, T, F, 0, F, xc1, 1, Fp)
    if train:
        assert ops.lstm_prefill_train(T, B, H, ndir, yp1, fw1, bw1)
    else:
        ops.lstm_prefill_fwd(T, B, ndir * H, yp1, fw1)
    yp2, fw2, bw2 = buffers()
    xc2 = torch.full((T, B, Fp), 5.0, device='cuda')
    assert ops.encoder_prologue(x, B, T, F, xc2, Fp, H, ndir, yp2, fw2, bw2)
    torch.cuda.synchronize()
    assert torch.equal(xc1, xc2)
    for a, b in zip(yp1 + fw1 + (bw1 or []), yp2 + fw2 + (bw2 or [])):
        assert torch.equal(a.view(torch.uint8).view(-1), b.view(torch.uint8).view(-1))
    assert int(yp2[0].view(torch.int32)[1, 0, 0]) == -1 and float(yp2[0][0].abs().max()) == 0.0


# ---------------------------------------------------------------- frontend
def test_frontend_and_reattach():
    from danet_amd import ops
    rng = np.random.RandomState(1)
    B, C, T, F = 3, 2, 7, 33
    src = (rng.randn(B, C, T, F) + 1j * rng.randn(B, C, T, F)).astype(np.complex64) * 50
    src[0, :, 0, :4] = 0                       # atan2(0,0) / log1p(0) / |0|
    fe = ops.frontend(torch.as_tensor(src).cuda(), want_phase=True, want_mix=True)
    ref = O.frontend(src)
    for k in ('src_pwr', 'mix_pwr', 'mix_log', 'phase', 'mix'):
        assert relerr(fe[k].cpu().numpy(), ref[k]) < 1e-5, k
    ph = fe['phasor'].cpu().numpy()
    assert relerr(ph[..., 0], np.cos(ref['phase'])) < 1e-5
    assert relerr(ph[..., 1], np.sin(ref['phase'])) < 1e-5
    pw = rng.rand(B, C, T, F).astype(np.float32)
    idx = np.array([1, 0, 1], dtype=np.int32)
    got = ops.reattach_phase(cu(pw), fe['phasor'], torch.as_tensor(idx).cuda()).cpu().numpy()
    perms = O.permutations(C)
    want = O.reattach_phase(O.perm_gather(pw.astype(np.float64), perms, idx), ref['phase'])
    assert relerr(got, want) < 1e-5


# ------------------------------------------------------------------- LSTM
def _lstm_ref(x, Ws, bs, H, dy):
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    Wt = [torch.tensor(W, dtype=torch.float64, requires_grad=True) for W in Ws]
    bt = [torch.tensor(b, dtype=torch.float64, requires_grad=True) for b in bs]
    outs = [R.lstm_scan(xt, Wt[0], bt[0], H)]
    if len(Ws) == 2:
        outs.append(R.lstm_scan(xt, Wt[1], bt[1], H, reverse=True))
    y = torch.cat(outs, dim=-1)
    (y * torch.tensor(dy)).sum().backward()
    return y.detach().numpy(), xt.grad.numpy(), [w.grad.numpy() for w in Wt], [b.grad.numpy() for b in bt]


@pytest.mark.parametrize('B,T,D,H,ndir', [
    (2, 5, 7, 4, 2), (1, 9, 5, 8, 1), (4, 16, 129, 12, 2), (17, 6, 20, 36, 2),
    (32, 12, 64, 300, 2), (33, 4, 16, 16, 2), (16, 8, 24, 600, 1)])
def test_lstm_layer_fwd_bwd(B, T, D, H, ndir):
    from danet_amd import ops
    rng = np.random.RandomState(B * 100 + T * 10 + H)
    r = 0.75 / np.sqrt(H)
    x = rng.randn(B, T, D) * 0.7
    Ws = [rng.uniform(-r, r, size=(D + H, 4 * H)) * 2 for _ in range(ndir)]
    bs = [O.lstm_bias_init(H) + rng.randn(4 * H) * 0.1 for _ in range(ndir)]
    dy = rng.randn(B, T, ndir * H)
    ry, rdx, rdW, rdb = _lstm_ref(x, Ws, bs, H, dy)
    xc = cu(x).requires_grad_(True)
    params = []
    for W, b in zip(Ws, bs):
        params += [cu(W).requires_grad_(True), cu(b).requires_grad_(True)]
    y = ops.LstmLayerFn.apply(xc, H, *params)
    assert relerr(y.detach().cpu().numpy(), ry) < TOL
    y.backward(cu(dy))
    assert relerr(xc.grad.cpu().numpy(), rdx) < TOL
    for d in range(ndir):
        assert relerr(params[2 * d].grad.cpu().numpy(), rdW[d]) < TOL
        assert relerr(params[2 * d + 1].grad.cpu().numpy(), rdb[d]) < TOL


def test_lstm_kat_gate_order_no_tanh():
    '''K1/K2: W=0, b=[1|1.5|-1|1] => c1=sigma(1.5), h1=sigma(1)tanh(c1),
    c2=sigma(1.5)+sigma(-1)c1; with the reference bias init (g bias 0) h == 0'''
    from danet_amd import ops
    H, D, B, T = 4, 3, 2, 2
    x = torch.randn(B, T, D, device='cuda')
    W = torch.zeros(D + H, 4 * H, device='cuda')
    b = cu(np.concatenate([np.full(H, 1.0), np.full(H, 1.5), np.full(H, -1.0), np.full(H, 1.0)]))
    y = ops.LstmLayerFn.apply(x, H, W, b).cpu().numpy()
    sig = lambda v: 1 / (1 + np.exp(-v))
    c1 = sig(1.5); h1 = sig(1.0) * np.tanh(c1)
    c2 = sig(1.5) + sig(-1.0) * c1; h2 = sig(1.0) * np.tanh(c2)
    assert np.allclose(y[:, 0], h1, atol=1e-6) and abs(h1 - 0.492549) < 1e-6
    assert np.allclose(y[:, 1], h2, atol=1e-6)
    y0 = ops.LstmLayerFn.apply(x, H, W, cu(O.lstm_bias_init(H))).cpu().numpy()
    assert np.all(y0 == 0.0)


# -------------------------------------------------------------- estimators
def _embed_case(rng, B, C, T, F, E):
    embed = rng.randn(B, T, F, E)
    src = (rng.randn(B, C, T, F) + 1j * rng.randn(B, C, T, F)) * 4
    src[0, :, 0] = 0                                       # argmax ties -> index 0 (K9)
    fe = O.frontend(src)
    return embed, fe['src_pwr'], fe['mix_pwr']


@pytest.mark.parametrize('mode', ['truth', 'truth-threshold', 'truth-weighted'])
@pytest.mark.parametrize('B,C,T,F,E', [(2, 2, 5, 7, 3), (3, 3, 40, 129, 20), (2, 2, 33, 65, 40),
                                       (8, 2, 40, 129, 20)])   # B % 8 == 0: the XCD-aware workgroup mapping
def test_truth_estimators(mode, B, C, T, F, E):
    from danet_amd import ops
    rng = np.random.RandomState(5)
    embed, src_pwr, mix_pwr = _embed_case(rng, B, C, T, F, E)
    ref = {'truth': O.est_truth, 'truth-threshold': O.est_truth_threshold,
           'truth-weighted': O.est_truth_weighted}[mode](embed, src_pwr, mix_pwr)
    e = cu(embed).requires_grad_(True)
    attr = ops.TruthAttractorFn.apply(e, cu(src_pwr), cu(mix_pwr), ops.TRUTH_MODES[mode], 1e-7)
    assert relerr(attr.detach().cpu().numpy(), ref) < TOL
    dattr = rng.randn(B, C, E)
    attr.backward(cu(dattr))
    et = torch.tensor(embed, requires_grad=True)
    fn = {'truth': R.est_truth, 'truth-threshold': R.est_truth_threshold,
          'truth-weighted': R.est_truth_weighted}[mode]
    (fn(et, torch.tensor(src_pwr), torch.tensor(mix_pwr), 1e-7) * torch.tensor(dattr)).sum().backward()
    assert relerr(e.grad.cpu().numpy(), et.grad.numpy()) < TOL


def test_truth_kats():
    '''K5: one bin => attr = embed/2 ('truth' divides by count+1);
    K6: |mix| == 5.0 excluded, next float above included'''
    from danet_amd import ops
    embed = cu(np.array([[[[2.0, 4.0, 6.0, 8.0]]]]))                      # B=T=F=1, E=4
    src_pwr = cu(np.array([[[[1.0]], [[0.5]]]]).reshape(1, 2, 1, 1))
    mix = cu(np.array([[[1.0]]]))
    a = ops.TruthAttractorFn.apply(embed, src_pwr, mix, 0, 1e-7).cpu().numpy()
    assert np.allclose(a[0, 0], [1, 2, 3, 4]) and np.all(a[0, 1] == 0)
    five = np.float32(5.0)
    mixv = np.array([[[five, np.nextafter(five, np.float32(10))]]], dtype=np.float32)   # [1,1,2]
    emb = cu(np.array([[[[1.0, 0, 0, 0], [0, 1.0, 0, 0]]]]))              # [1,1,2,4]
    sp = cu(np.ones((1, 2, 1, 2))); sp[0, 1] = 0.5
    a = ops.TruthAttractorFn.apply(emb, sp, cu(mixv), 1, 1e-7).cpu().numpy()
    assert abs(a[0, 0, 0]) < 1e-6 and abs(a[0, 0, 1] - 1.0) < 1e-5


@pytest.mark.parametrize('B,C,T,F,E,A', [(2, 2, 5, 7, 3, 4), (3, 2, 40, 129, 20, 6),
                                         (2, 3, 17, 65, 20, 5), (1, 2, 300, 129, 20, 6),
                                         (8, 2, 40, 129, 20, 6), (16, 2, 17, 129, 20, 6)])   # B % 8 == 0: XCD-aware mapping
def test_anchor_estimator(B, C, T, F, E, A):
    from danet_amd import ops
    rng = np.random.RandomState(11)
    embed = rng.randn(B, T, F, E) * 0.5
    anchors = rng.randn(A, E)
    ref, info = O.est_anchor(embed, anchors, C, return_all=True)
    e = cu(embed).requires_grad_(True)
    an = cu(anchors).requires_grad_(True)
    attr, asets, choice = ops.AnchorAttractorFn.apply(e, an, C)
    assert np.array_equal(choice.cpu().numpy(), info['subset_choice'])
    assert relerr(asets.cpu().numpy(), info['asets']) < TOL
    assert relerr(attr.detach().cpu().numpy(), ref) < TOL
    dattr = rng.randn(B, C, E)
    attr.backward(cu(dattr))
    et = torch.tensor(embed, requires_grad=True)
    at = torch.tensor(anchors, requires_grad=True)
    (R.est_anchor(et, at, C) * torch.tensor(dattr)).sum().backward()
    assert relerr(e.grad.cpu().numpy(), et.grad.numpy()) < TOL
    assert relerr(an.grad.cpu().numpy(), at.grad.numpy()) < TOL


def test_anchor_kat_diagonal_in_max():
    '''K7: the similarity max runs over the full CxC Gram matrix, diagonal
    included: a subset with a long attractor loses even if its attractors are
    orthogonal'''
    from danet_amd import ops
    rng = np.random.RandomState(3)
    embed = rng.randn(1, 6, 9, 3) * np.array([3.0, 0.2, 0.2])
    anchors = np.array([[4.0, 0, 0], [-4.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]])
    ref, info = O.est_anchor(embed, anchors, 2, return_all=True)
    gram = info['asets'] @ np.swapaxes(info['asets'], -1, -2)
    offdiag = np.array([[g[0, 1] for g in gb] for gb in gram])
    assert np.argmin(offdiag[0]) != info['subset_choice'][0]   # the KAT is discriminating
    _, _, choice = ops.AnchorAttractorFn.apply(cu(embed), cu(anchors), 2)
    assert int(choice[0]) == int(info['subset_choice'][0])


# --------------------------------------------------------------- separators
@pytest.mark.parametrize('act', [0, 1])
@pytest.mark.parametrize('B,C,T,F,E', [(2, 2, 5, 7, 3), (3, 3, 40, 129, 20), (2, 2, 20, 65, 40), (8, 2, 40, 129, 20)])
def test_separators(act, B, C, T, F, E):
    from danet_amd import ops
    rng = np.random.RandomState(7)
    embed = rng.randn(B, T * F, E)
    attr = rng.randn(B, C, E)
    mix = rng.rand(B, T, F) * 10
    name = 'softmax' if act == 0 else 'sigmoid'
    ref, rmask = O.sep_dot(mix, attr, embed, name, return_masks=True)
    e = cu(embed).requires_grad_(True)
    a = cu(attr).requires_grad_(True)
    sep, masks = ops.SeparateFn.apply(cu(mix), a, e, act, True)
    assert relerr(sep.detach().cpu().numpy(), ref) < TOL
    assert relerr(masks.cpu().numpy(), rmask) < TOL
    dout = rng.randn(B, C, T, F)
    sep.backward(cu(dout))
    et = torch.tensor(embed, requires_grad=True)
    at = torch.tensor(attr, requires_grad=True)
    (R.sep_dot(torch.tensor(mix), at, et, name)[0] * torch.tensor(dout)).sum().backward()
    assert relerr(e.grad.cpu().numpy(), et.grad.numpy()) < TOL
    assert relerr(a.grad.cpu().numpy(), at.grad.numpy()) < TOL


def test_separator_kat_equal_attractors():
    '''K4: equal attractors => softmax masks 1/C; zero logits => sigmoid 0.5'''
    from danet_amd import ops
    B, C, T, F, E = 1, 3, 2, 5, 4
    embed = torch.randn(B, T * F, E, device='cuda')
    attr = torch.randn(B, 1, E, device='cuda').expand(B, C, E).contiguous()
    mix = torch.rand(B, T, F, device='cuda') + 1
    sep, _ = ops.SeparateFn.apply(mix, attr, embed, 0, False)
    assert torch.allclose(sep, (mix / C)[:, None].expand(B, C, T, F), rtol=1e-6)
    sep, _ = ops.SeparateFn.apply(mix, torch.zeros(B, C, E, device='cuda'), embed, 1, False)
    assert torch.allclose(sep, (mix * 0.5)[:, None].expand(B, C, T, F), rtol=1e-6)


# --------------------------------------------------------------------- loss
@pytest.mark.parametrize('mode', [0, 1])
@pytest.mark.parametrize('B,C,T,F', [(2, 2, 5, 7), (5, 3, 40, 129), (32, 2, 16, 129), (3, 1, 4, 9)])
def test_pit_mse(mode, B, C, T, F):
    from danet_amd import ops
    rng = np.random.RandomState(13)
    src = (rng.randn(B, C, T, F) + 1j * rng.randn(B, C, T, F)) * 3
    fe = O.frontend(src)
    # estimates close to a random permutation of the truth so every perm is hit
    pick = rng.randint(0, len(O.permutations(C)), size=B)
    sep = np.abs(src)[np.arange(B)[:, None], O.permutations(C)[pick]] * (0.8 + 0.4 * rng.rand(B, C, T, F))
    phasor = np.stack([np.cos(fe['phase']), np.sin(fe['phase'])], -1)
    if mode == 0:
        est = O.reattach_phase(sep, fe['phase'])
        rloss, perms, ridx, _ = O.pit_mse_loss(src, est)
    else:
        rloss, perms, ridx, _ = O.pit_mse_loss(fe['src_pwr'], sep)
    rsnr = O.batch_snr(src, O.reattach_phase(O.perm_gather(sep, perms, ridx), fe['phase'])).mean()
    s = cu(sep).requires_grad_(True)
    loss, snr, idx = ops.PitMseFn.apply(torch.as_tensor(src.astype(np.complex64)).cuda(), s,
                                        cu(phasor), mode, 1e-7)
    assert np.array_equal(idx.cpu().numpy(), ridx)
    assert relerr(float(loss.detach()), rloss) < TOL
    assert relerr(float(snr), rsnr) < TOL
    (loss * 1.7).backward()
    st = torch.tensor(sep, requires_grad=True)
    if mode == 0:
        ph = torch.tensor(fe['phase'])[:, None]
        est = torch.complex(torch.cos(ph) * st, torch.sin(ph) * st)
        tl = R.pit_mse_loss(torch.tensor(src), est)[0]
    else:
        tl = R.pit_mse_loss(torch.tensor(fe['src_pwr']), st)[0]
    (tl * 1.7).backward()
    assert relerr(s.grad.cpu().numpy(), st.grad.numpy()) < TOL


def test_pit_kat_swap_and_order():
    '''K8: swapping the sources flips idx 0 -> 1 with the same loss; 3-speaker
    permutations follow itertools.permutations order'''
    from danet_amd import ops
    rng = np.random.RandomState(2)
    B, C, T, F = 1, 2, 3, 5
    src = (rng.randn(B, C, T, F) + 1j * rng.randn(B, C, T, F)).astype(np.complex64)
    fe = O.frontend(src)
    phasor = cu(np.stack([np.cos(fe['phase']), np.sin(fe['phase'])], -1))
    sep = np.abs(src).astype(np.float32)
    l0, _, i0 = ops.PitMseFn.apply(torch.as_tensor(src).cuda(), cu(sep), phasor, 1, 1e-7)
    l1, _, i1 = ops.PitMseFn.apply(torch.as_tensor(src).cuda(), cu(sep[:, ::-1].copy()), phasor, 1, 1e-7)
    assert int(i0[0]) == 0 and int(i1[0]) == 1 and abs(float(l0) - float(l1)) < 1e-7
    C = 3
    src = ((rng.randn(1, C, T, F) + 1j * rng.randn(1, C, T, F))
           * np.array([1, 5, 25]).reshape(1, 3, 1, 1)).astype(np.complex64)
    fe = O.frontend(src)
    phasor = cu(np.stack([np.cos(fe['phase']), np.sin(fe['phase'])], -1))
    for p, perm in enumerate(itertools.permutations(range(C))):
        # estimate j = |src_i| for perm[i] = j
        sep = np.zeros((1, C, T, F), np.float32)
        for i, j in enumerate(perm):
            sep[:, j] = np.abs(src[:, i])
        _, _, idx = ops.PitMseFn.apply(torch.as_tensor(src).cuda(), cu(sep), phasor, 1, 1e-7)
        assert int(idx[0]) == p


# ---------------------------------------------------------------- STFT/iSTFT
def test_stft_against_reference_golden(hp):
    from danet_amd import ops
    g = np.load(GOLD)
    w256 = O.fft_window(256)
    assert np.array_equal(w256.view(np.uint32), g['wnd256_bits'])            # G1
    for L in (256, 257, 319, 320, 8000, 8001, 8063, 8064):                   # G2
        x = np.random.RandomState(0).randn(L).astype(np.float32)
        X = ops.stft(cu(x), cu(w256), 256, 64).cpu().numpy()
        ref = g['stft256_L%d' % L]
        assert X.shape == ref.shape == (1 + -(-L // 64), 129)                # K10, exact
        assert relerr(X, ref) < 1e-5
    X = ops.stft(cu(g['stft256_int16scale_x']), cu(w256), 256, 64).cpu().numpy()
    assert relerr(X, g['stft256_int16scale']) < 1e-5
    with pytest.raises(ValueError):                                           # K10: Ls < N
        ops.stft(cu(np.zeros(255, np.float32)), cu(w256), 256, 64)
    assert int(g['stft256_short_raises']) == 1
    # batched call == per-signal calls
    xs = np.random.RandomState(4).randn(3, 1000).astype(np.float32)
    Xb = ops.stft(cu(xs), cu(w256), 256, 64).cpu().numpy()
    for i in range(3):
        assert relerr(Xb[i], O.stft(xs[i], w256, 256, 64)) < 1e-5


def test_wav_io_against_reference_golden(hp, tmp_path):
    '''G6 / G7: the demo path's wav I/O -- utils.load_wavfile (scipy FFT resampling to SMPRATE
    with the reference's ceil'ed length, then the HIP STFT) and utils.save_wavfile (HIP iSTFT, wav
    of the same dtype / rate) against what the reference's own app/utils.py:95-135 returned for
    the same files (tests/golden/make_golden.py)'''
    import scipy.io.wavfile
    from danet_amd import utils
    hp.load(dict(FFT_SIZE=256, FFT_STRIDE=64, SMPRATE=8000))
    hp.digest()
    g = np.load(GOLD)
    for rate in (8000, 11025, 16000):
        fn = str(tmp_path / ('in_%d.wav' % rate))
        scipy.io.wavfile.write(fn, rate, g['wav_in_%d' % rate])
        X = utils.load_wavfile(fn)
        ref = g['wav_load_%d' % rate]
        assert X.shape == ref.shape and X.dtype == ref.dtype
        assert relerr(X, ref) < 1e-5
    fn = str(tmp_path / 'out.wav')
    utils.save_wavfile(fn, g['wav_load_11025'])
    sr, back = scipy.io.wavfile.read(fn)
    assert sr == int(g['wav_save_rate']) and str(back.dtype) == str(g['wav_save_dtype'])
    assert back.shape == g['wav_save_data'].shape
    assert relerr(back, g['wav_save_data']) < 1e-5
    with pytest.raises(IOError):
        utils.load_wavfile(None)


def test_stft_512_long_utterance_golden():
    from danet_amd import ops
    g = np.load(GOLD)
    w = O.fft_window(512)
    assert np.array_equal(w.view(np.uint32), g['wnd512_bits'])
    x = np.random.RandomState(0).randn(160000).astype(np.float32)
    X = ops.stft(cu(x), cu(w), 512, 128)
    assert tuple(X.shape) == tuple(g['stft512_L160000_shape']) == (1251, 257)
    Xn = X.cpu().numpy()
    scale = np.abs(Xn).max()
    assert np.abs(Xn[:2] - g['stft512_L160000_head']).max() / scale < 1e-5
    assert np.abs(Xn[-2:] - g['stft512_L160000_tail']).max() / scale < 1e-5
    assert np.abs(Xn[600:602] - g['stft512_L160000_mid']).max() / scale < 1e-5
    assert abs(np.abs(Xn.astype(np.complex128)).sum() - g['stft512_L160000_abs_sum']) \
        < 1e-5 * g['stft512_L160000_abs_sum']
    y = ops.istft(X, 128, cu(w)).cpu().numpy()
    assert len(y) == int(g['istft512_L160000_len'])
    sc = np.abs(g['istft512_L160000_head']).max()
    assert np.abs(y[:1024] - g['istft512_L160000_head']).max() / sc < 1e-5
    assert np.abs(y[-1024:] - g['istft512_L160000_tail']).max() / sc < 1e-5
    assert abs(np.abs(y).sum() - g['istft512_L160000_abs_sum']) < 1e-5 * g['istft512_L160000_abs_sum']


def test_istft_against_reference_golden():
    from danet_amd import ops
    g = np.load(GOLD)
    w = O.fft_window(256)
    for L in (256, 257, 320, 8000, 8064):                                    # G3
        X = g['stft256_L%d' % L]
        ref = g['istft256_L%d' % L]
        y = ops.istft(torch.as_tensor(X).cuda(), 64, cu(w)).cpu().numpy()
        assert y.shape == ref.shape and y.dtype == np.float64
        assert np.abs(y - ref).max() <= 1e-5 * max(np.abs(ref).max(), 1e-30)
        # frame truncation: the tail the reference never writes stays exactly 0
        used = O.istft_num_frames_used(X.shape[0], 256, 64)
        assert np.all(y[(used - 1) * 64 + 256:] == 0) if used > 0 else np.all(y == 0)


def test_stft_istft_round_trip_property():
    '''size-independent property: istft(stft(x)) * sum(w) == x on the samples
    the reference's istft covers with full window overlap'''
    from danet_amd import ops
    w = O.fft_window(256)
    x = np.random.RandomState(9).randn(4, 40000).astype(np.float32)
    X = ops.stft(cu(x), cu(w), 256, 64)
    y = ops.istft(X, 64, cu(w)).cpu().numpy() * float(w.astype(np.float64).sum())
    # stft sample i sits at extended index i+128; covered region: full overlap
    T = X.shape[1]
    used = O.istft_num_frames_used(T, 256, 64)
    lo, hi = 256, (used - 1) * 64
    assert np.abs(y[:, lo:hi] - x[:, lo - 128:hi - 128]).max() < 2e-4 * np.abs(x).max()


# --------------------------------------------------------------- whole model
def _model_case(hp, cfg, B, C, T, seed=0):
    from danet_amd.model import Model
    hp.load(dict(BATCH_SIZE=B, MAX_N_SIGNAL=C, FFT_SIZE=cfg['FFT'], FFT_STRIDE=cfg['FFT'] // 4,
                 EMBED_SIZE=cfg['E'], NUM_LSTM_LAYERS=cfg['L'], LSTM_HDIM=cfg['H'],
                 NUM_ANCHOR=cfg['A'], ENCODER_TYPE=cfg.get('encoder', 'bilstm-orig'),
                 TRAIN_ESTIMATOR_METHOD=cfg['train_est'],
                 INFER_ESTIMATOR_METHOD=cfg['infer_est'], SEPARATOR_TYPE=cfg['separator']))
    hp.digest()
    F = hp.FEATURE_SIZE
    rng = np.random.RandomState(seed)
    src = ((rng.randn(B, C, T, F) + 1j * rng.randn(B, C, T, F)) * 6).astype(np.complex64)
    model = Model('t', device='cuda').build()
    return model, src


@pytest.mark.parametrize('train_est,sepn', [('truth-weighted', 'dot-sigmoid-orig'),
                                            ('anchor', 'dot-softmax-orig'),
                                            ('truth', 'dot-softmax-orig'),
                                            ('truth-threshold', 'dot-sigmoid-orig')])
def test_model_forward_backward_tiny(hp, train_est, sepn):
    '''G5-style: every debug_fetches intermediate + loss/SNR/perm + all
    parameter gradients vs the oracle at B=2,C=2,T=8,F=5,E=3,H=4,L=2,A=4'''
    cfg = dict(FFT=8, H=4, L=2, E=3, C=2, A=4, train_est=train_est, infer_est='anchor',
               separator=sepn, with_valid=True)
    model, src = _model_case(hp, cfg, B=2, C=2, T=8)
    params = model.param_dict()
    model._flat_grad.zero_()
    out = model.forward(torch.as_tensor(src).cuda(), with_valid=True)
    out['loss'].backward()
    ref = O.model_forward(src.astype(np.complex128), params, cfg)
    for k in ('embed', 'attrs', 'sep_pwr', 'valid_attrs', 'sep_pwr_valid'):
        assert relerr(out[k].detach().cpu().numpy(), ref[k]) < TOL, k
    for k in ('loss', 'SNR', 'valid_loss', 'valid_SNR'):
        assert relerr(float(out[k]), ref[k]) < TOL, k
    assert np.array_equal(out['perm_idx'].cpu().numpy(), ref['perm_idx'])
    assert np.array_equal(out['valid_perm_idx'].cpu().numpy(), ref['valid_perm_idx'])
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params.items()}
    r = R.model_forward(torch.tensor(src.astype(np.complex128)), tp, cfg)
    r['loss'].backward()
    g = model.grad_dict()
    for k in params:
        if tp[k].grad is None:
            assert np.all(g[k] == 0), k                    # e.g. unused infer anchors
        else:
            assert relerr(g[k], tp[k].grad.numpy()) < 2 * TOL, k


def test_model_cfg1_toy_data(hp):
    '''BASELINE cfg 1: toy white-noise batch (reference generator), B=4, C=2,
    F=129, 1-layer BiLSTM H=300, truth-weighted + anchor, softmax separator'''
    cfg = dict(FFT=256, H=300, L=1, E=20, C=2, A=6, train_est='truth-weighted',
               infer_est='anchor', separator='dot-softmax-orig', with_valid=True)
    model, _ = _model_case(hp, cfg, B=4, C=2, T=4)
    src = O.toy_batch(np.random.RandomState(1337), 4, 2, hp.FEATURE_SIZE, T=128)
    params = model.param_dict()
    model._flat_grad.zero_()
    out = model.forward(torch.as_tensor(src).cuda(), with_valid=True)
    out['loss'].backward()
    ref = O.model_forward(src.astype(np.complex128), params, cfg)
    for k in ('embed', 'attrs', 'sep_pwr', 'sep_pwr_valid'):
        assert relerr(out[k].detach().cpu().numpy(), ref[k]) < TOL, k
    for k in ('loss', 'SNR', 'valid_loss', 'valid_SNR'):
        assert relerr(float(out[k]), ref[k]) < TOL, k
    assert np.array_equal(out['valid_perm_idx'].cpu().numpy(), ref['valid_perm_idx'])
    mask_mse = float(np.mean((out['sep_pwr'].detach().cpu().numpy() - ref['sep_pwr']) ** 2))
    assert mask_mse < 1e-8 * float(np.mean(ref['sep_pwr'] ** 2)) + 1e-12


def test_train_step_matches_tf_adam(hp):
    '''two optimiser steps == oracle TF1-Adam (eps outside the root) + clip'''
    cfg = dict(FFT=8, H=4, L=1, E=3, C=2, A=4, train_est='truth-weighted', infer_est='anchor',
               separator='dot-sigmoid-orig')
    model, src = _model_case(hp, cfg, B=2, C=2, T=6)
    p0 = model.param_dict()
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p0.items()}
    m = {k: torch.zeros_like(v) for k, v in tp.items()}
    v = {k: torch.zeros_like(v_) for k, v_ in tp.items()}
    for t in (1, 2):
        model.train_step(torch.as_tensor(src).cuda())
        for k in tp:
            tp[k].grad = None
        R.model_forward(torch.tensor(src.astype(np.complex128)), tp, cfg)['loss'].backward()
        R.tf_adam_step_(tp, {k: tp[k].grad for k in tp}, m, v, t, hp.LR, clip=hp.GRAD_CLIP_THRES)
    p2 = model.param_dict()
    for k in p0:
        moved = np.abs(tp[k].detach().numpy() - p0[k]).max()
        assert np.abs(p2[k] - tp[k].detach().numpy()).max() <= 2e-3 * moved + 1e-7, k


def test_lstm_orig_encoder_and_lyr_lstm_api(hp):
    '''unidirectional `lstm-orig` plugin and the Model.lyr_lstm entry point'''
    cfg = dict(FFT=8, H=8, L=2, E=3, C=2, A=4, train_est='truth-weighted', infer_est='anchor',
               separator='dot-sigmoid-orig', encoder='lstm-orig')
    model, src = _model_case(hp, cfg, B=2, C=2, T=7)
    params = model.param_dict()
    out = model.forward(torch.as_tensor(src).cuda())
    ref = O.model_forward(src.astype(np.complex128), params, cfg)
    assert relerr(out['embed'].detach().cpu().numpy(), ref['embed']) < TOL
    assert relerr(float(out['loss']), ref['loss']) < TOL


def test_infer_and_debug_fetches(hp):
    cfg = dict(FFT=8, H=4, L=1, E=3, C=2, A=4, train_est='truth-weighted', infer_est='anchor',
               separator='dot-softmax-orig')
    hp.load(dict(DEBUG=True))
    model, src = _model_case(hp, cfg, B=2, C=2, T=8)
    params = model.param_dict()
    mix = src.sum(axis=1)
    got = model.infer(torch.as_tensor(mix).cuda()).cpu().numpy()
    ref = O.model_forward(src.astype(np.complex128), params, dict(cfg, with_valid=True))
    assert relerr(got, ref['signals_infer']) < TOL
    dbg = model.debug_fetch(torch.as_tensor(src).cuda())
    assert relerr(dbg['output'].cpu().numpy(), ref['output']) < TOL
    assert relerr(dbg['masks'].cpu().numpy(), ref['masks']) < TOL


@pytest.mark.parametrize('B,T,D,H,ndir', [
    (64, 6, 32, 300, 2),      # fwd: 32-row clusters (MT=2); bwd: 16-row clusters
    (96, 3, 16, 300, 2),
    (8, 5, 16, 600, 2),       # H=600 bidirectional: 150/76 workgroups, 154 KB LDS in BPTT
    (3, 1, 9, 8, 2),          # T=1: no recurrent step at all
    (40, 4, 12, 64, 1),
])
def test_lstm_layer_shape_envelope(B, T, D, H, ndir):
    '''shapes outside the BASELINE configs that `-bs` / hparams can reach'''
    from danet_amd import ops
    rng = np.random.RandomState(B + T + H)
    r = 0.75 / np.sqrt(H)
    x = rng.randn(B, T, D) * 0.7
    Ws = [rng.uniform(-r, r, size=(D + H, 4 * H)) * 2 for _ in range(ndir)]
    bs = [O.lstm_bias_init(H) + rng.randn(4 * H) * 0.1 for _ in range(ndir)]
    dy = rng.randn(B, T, ndir * H)
    ry, rdx, rdW, rdb = _lstm_ref(x, Ws, bs, H, dy)
    xc = cu(x).requires_grad_(True)
    params = []
    for W, b in zip(Ws, bs):
        params += [cu(W).requires_grad_(True), cu(b).requires_grad_(True)]
    y = ops.LstmLayerFn.apply(xc, H, *params)
    assert relerr(y.detach().cpu().numpy(), ry) < TOL
    y.backward(cu(dy))
    assert relerr(xc.grad.cpu().numpy(), rdx) < TOL
    for d in range(ndir):
        assert relerr(params[2 * d].grad.cpu().numpy(), rdW[d]) < TOL
        assert relerr(params[2 * d + 1].grad.cpu().numpy(), rdb[d]) < TOL


@pytest.mark.parametrize('B,T,D,H,ndir,force', [
    (32, 6, 40, 600, 2, None),     # cfg 4 as written: chosen by itself (200 workgroups of 16 rows)
    (20, 5, 16, 36, 2, '12'),      # 3 groups of 12 units, ragged batch
    (33, 4, 12, 40, 2, '12'),      # last group has 4 of 12 units, 3 row clusters
    (16, 7, 24, 132, 1, '12'),     # 11 groups, one direction
    (32, 6, 40, 600, 2, '8'),      # and the 8-unit / 32-row geometry it replaces
])
def test_lstm_forward_12_units_per_workgroup(B, T, D, H, ndir, force, monkeypatch):
    '''lstm_fwd_kernel<1, 4, 12>: 48 gate columns (3 MFMA column tiles) per workgroup -- the
    geometry wide layers use so that 16-row clusters fit the GPU -- against the oracle, forward and
    (through the unchanged BPTT) backward.  The fused forward is off: this is the hoisted kernel.'''
    from danet_amd import ops
    from danet_amd import _lib
    _lib.set_option('lstm_fwd_fused', 0)
    if force:
        _lib.set_option('lstm_fwd_un', int(force))
    rng = np.random.RandomState(B + 7 * T + H)
    r = 0.75 / np.sqrt(H)
    x = rng.randn(B, T, D) * 0.7
    Ws = [rng.uniform(-r, r, size=(D + H, 4 * H)) * 2 for _ in range(ndir)]
    bs = [O.lstm_bias_init(H) + rng.randn(4 * H) * 0.1 for _ in range(ndir)]
    dy = rng.randn(B, T, ndir * H)
    ry, rdx, rdW, rdb = _lstm_ref(x, Ws, bs, H, dy)
    xc = cu(x).requires_grad_(True)
    params = []
    for W, b in zip(Ws, bs):
        params += [cu(W).requires_grad_(True), cu(b).requires_grad_(True)]
    y = ops.LstmLayerFn.apply(xc, H, *params)
    assert relerr(y.detach().cpu().numpy(), ry) < TOL
    y.backward(cu(dy))
    assert relerr(xc.grad.cpu().numpy(), rdx) < TOL
    for d in range(ndir):
        assert relerr(params[2 * d].grad.cpu().numpy(), rdW[d]) < TOL


def test_lstm_unsupported_shapes_fail_loudly():
    '''outside the compiled envelope the library refuses (no silent fallback)'''
    from danet_amd import ops, _lib
    x = torch.randn(2, 3, 5, device='cuda')
    H = 6                                           # not a multiple of 4
    W = torch.randn(5 + H, 4 * H, device='cuda')
    b = torch.zeros(4 * H, device='cuda')
    with pytest.raises(_lib.DanetHipError):
        ops.LstmLayerFn.apply(x, H, W, b)


@pytest.mark.parametrize('B,T,H', [(2, 96, 300), (5, 40, 36), (16, 50, 128), (32, 128, 300),
                                   (48, 33, 300), (32, 24, 600), (7, 25, 20), (64, 16, 300)])
def test_lstm_bwd_handoff_under_uneven_load(B, T, H):
    '''the BPTT kernel's inter-workgroup hand-off (phase-tagged partial-dh ring,
    csrc/lstm.hip) repeated under an UNEVEN concurrent load (a GEMM stream that comes
    and goes on another HIP stream), every output word compared with an undisturbed launch
    (the kernel vs the float64 oracle: test_lstm_layer_fwd_bwd, test_bptt_vs_oracle_long_sequence)
    and the hand-off status word checked'''
    from danet_amd import _lib
    L = _lib.load()
    ptr = _lib.ptr
    dev = torch.device('cuda')
    g = torch.Generator(device='cuda').manual_seed(B * 1000 + T * 10 + H)
    rnd = lambda *s: torch.randn(*s, device=dev, generator=g)
    gx = [rnd(T * B, 4 * H) * 0.5 for _ in range(2)]
    Wh = [rnd(H, 4 * H) * (0.75 / H ** 0.5) for _ in range(2)]
    dy = rnd(T, B, 2 * H)
    n = _lib.ws_bytes(_lib.WS_LSTM, T, B, H, 2)
    st = torch.cuda.current_stream().cuda_stream
    ypad = torch.empty(T + 2, B, 2 * H, device=dev)
    gates = [x.clone() for x in gx]
    cells = [torch.empty(T * B, H, device=dev) for _ in range(2)]
    ws = torch.zeros(n, dtype=torch.uint8, device=dev)
    _lib.check(L.danet_lstm_fwd(st, T, B, H, 2, ptr(gates[0]), ptr(gates[1]), ptr(Wh[0]), ptr(Wh[1]),
                                4 * H, ptr(ypad), 2 * H, ptr(gates[0]), ptr(gates[1]),
                                ptr(cells[0]), ptr(cells[1]), ptr(ws), n, None, 0))
    torch.cuda.synchronize()
    assert int(ws[:4].view(torch.int32)[0]) == 0

    def bwd():
        das = [torch.full((T * B, 4 * H), float('nan'), device=dev) for _ in range(2)]
        w = torch.zeros(n, dtype=torch.uint8, device=dev)
        _lib.check(L.danet_lstm_bwd(st, T, B, H, 2, ptr(dy), 2 * H, ptr(Wh[0]), ptr(Wh[1]), 4 * H,
                                    ptr(gates[0]), ptr(gates[1]), ptr(cells[0]), ptr(cells[1]),
                                    ptr(das[0]), ptr(das[1]), None, None, 0.0, ptr(w), n, None, 0))
        return das, w

    ref, w = bwd()                                 # an undisturbed launch as the reference
    torch.cuda.synchronize()
    assert int(w[:4].view(torch.int32)[0]) == 0
    scale = max(float(r.abs().max()) for r in ref)
    side = torch.cuda.Stream()
    a = torch.randn(2048, 2048, device=dev)
    for it in range(6):
        with torch.cuda.stream(side):            # bursts of different length: uneven load
            for _ in range(it * 3):
                a @ a
        das, w = bwd()
        torch.cuda.synchronize()
        assert int(w[:4].view(torch.int32)[0]) == 0, 'hand-off timeout (iteration %d)' % it
        for d in range(2):
            assert torch.isfinite(das[d]).all()
            assert float((das[d] - ref[d]).abs().max()) < 2e-5 * scale
