'''
fp32 products on the bf16 matrix cores (csrc/gemm_x6.hip; run with -m gpu), through the C ABI:
`danet_gemm_pack_weights` + `danet_gemm_x6` against the float64 product of the SAME fp32 operands,
with the exact-fp32 kernel's error beside it, and the host side that keeps the packed weights
current (ops.packed_weight / weights_written / repack_weights).

Tolerance: 4e-6 of the result's largest magnitude (an fp32 FMA chain over K = 2580 is at 0.7-3.5e-6
on these operands; the test also demands <= 3x the exact-fp32 kernel's own error).
'''
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 4e-6


def _err(C, ref):
    return float((C.double() - ref).abs().max() / ref.abs().max())


def _operands(M, N, K, seed, wide=False):
    g = torch.Generator(device='cuda').manual_seed(seed)
    A = torch.randn(M, K, device='cuda', generator=g)
    if wide:         # magnitudes over 12 binades, signs mixed: exercises the mid / lo pieces
        A = A * torch.exp2(torch.randint(-6, 6, (M, K), device='cuda', generator=g).float())
    W = (torch.rand(N, K, device='cuda', generator=g) - 0.5) * 0.2
    return A, W


# (M, N, K1, K2): the step's shapes at cfg 2 (projection, dYc, dX), cfg 4 at H = 600, ragged edges,
# a K that is not a multiple of 16, a single k-step, a product that is cut along K
SHAPES = [(4096, 2580, 600, 0), (4096, 600, 2580, 0), (4096, 600, 1200, 1200), (512, 1200, 2400, 2400),
          (257, 129, 20, 44), (130, 131, 36, 0), (64, 40, 8, 0), (1, 1, 4, 0), (1000, 600, 4100, 0)]


@pytest.mark.parametrize('M,N,K1,K2', SHAPES)
def test_x6_product_vs_float64(M, N, K1, K2):
    from danet_amd import ops
    A1, W1 = _operands(M, N, K1, 11 + M + N + K1, wide=True)
    ref = A1.double() @ W1.double().t()
    A2 = W2 = None
    if K2:
        A2, W2 = _operands(M, N, K2, 5 + M + K2)
        ref = ref + A2.double() @ W2.double().t()
    C = torch.full((M, N), float('nan'), device='cuda')
    assert ops._x6_ok(M, N, (A1, K1, K1))
    ops.gemm_w(A1, K1, W1, K1, 1, C, M, N, K1, N, A2=A2, lda2=K2, W2=W2, K2=K2, sn2=K2)
    C32 = torch.empty(M, N, device='cuda')
    ops.gemm(A1, W1, C32, M, N, K1, K1, K1, N, transB=True)
    if K2:
        ops.gemm(A2, W2, C32, M, N, K2, K2, K2, N, transB=True, beta=1.0)
    e6, e32 = _err(C, ref), _err(C32, ref)
    assert torch.isfinite(C).all()
    assert e6 <= TOL and e6 <= 3 * e32 + 1e-7, (e6, e32)
    # bit-reproducible (the K slices are summed in slice order)
    C2 = torch.empty_like(C)
    ops.gemm_w(A1, K1, W1, K1, 1, C2, M, N, K1, N, A2=A2, lda2=K2, W2=W2, K2=K2, sn2=K2)
    assert torch.equal(C, C2)


def test_x6_weight_given_k_major_and_padded_output_rows():
    '''the projection reads its weight as B(n, k) = W[k][n] (stride_n = 1) and the result may have
    ldc > N'''
    from danet_amd import ops
    M, N, K = 300, 260, 600
    g = torch.Generator(device='cuda').manual_seed(3)
    A = torch.randn(M, K, device='cuda', generator=g)
    W = torch.randn(K, N, device='cuda', generator=g) * 0.05
    C = torch.zeros(M, N + 12, device='cuda')
    ops.gemm_w(A, K, W, 1, N, C, M, N, K, N + 12)
    ref = A.double() @ W.double()
    assert _err(C[:, :N], ref) <= TOL
    assert float(C[:, N:].abs().max()) == 0.0               # nothing written beyond N


def test_x6_split_is_exact_and_special_values_propagate():
    '''hi + mid + lo == x exactly: a product with a one-hot weight returns A's columns bit for bit;
    zeros stay zeros, an inf / nan in A reaches exactly the rows it belongs to'''
    from danet_amd import ops
    M, N, K = 96, 64, 64
    g = torch.Generator(device='cuda').manual_seed(9)
    A = torch.randn(M, K, device='cuda', generator=g) * torch.exp2(
        torch.randint(-20, 20, (M, K), device='cuda', generator=g).float())
    W = torch.zeros(N, K, device='cuda')
    W[torch.arange(N), torch.arange(N)] = 1.0
    C = torch.empty(M, N, device='cuda')
    ops.gemm_w(A, K, W, K, 1, C, M, N, K, N)
    assert torch.equal(C, A[:, :N])
    A[5, 7] = float('inf')
    A[9, 3] = float('nan')
    ops.gemm_w(A, K, W, K, 1, C, M, N, K, N)
    bad = ~torch.isfinite(C)
    assert bad[5].any() and bad[9].any() and not bad[[0, 1, 2, 3, 4, 6, 7, 8] + list(range(10, M))].any()


def test_x6_unsupported_operands_are_refused_not_miscomputed():
    from danet_amd import ops, _lib
    A = torch.randn(64, 30, device='cuda')                  # K % 4 != 0
    assert not ops._x6_ok(64, 32, (A, 30, 30))
    W = torch.randn(32, 30, device='cuda')
    C = torch.empty(64, 32, device='cuda')
    with pytest.raises(_lib.DanetHipError):
        ops.gemm_w(A, 30, W, 30, 1, C, 64, 32, 30, 32)
    A = torch.randn(64 * 32 + 1, device='cuda')[1:].reshape(64, 32)     # 4-byte aligned only
    assert not ops._x6_ok(64, 32, (A, 32, 32))


def test_packed_weights_follow_the_parameters():
    '''a pack is refreshed when torch writes the weight (version counter), when a kernel of the
    library does (weights_written, called by adam_clip_step) and by repack_weights; an untouched
    weight is not packed again'''
    from danet_amd import ops
    M, N, K = 128, 128, 64
    g = torch.Generator(device='cuda').manual_seed(1)
    A = torch.randn(M, K, device='cuda', generator=g)
    theta = torch.randn(2 * N * K, device='cuda', generator=g)
    W = theta[N * K:].view(N, K)                            # a view into a flat buffer, as Model's
    C = torch.empty(M, N, device='cuda')

    def run():
        ops.gemm_w(A, K, W, K, 1, C, M, N, K, N)
        return _err(C, A.double() @ W.double().t())
    assert run() <= TOL
    pw = ops._packs[ops._dev_key(W.device)][(W.data_ptr(), N, K, K, 1)]
    assert not pw.stale
    W.mul_(2.0)                                             # torch-side write
    assert run() <= TOL
    grad, m, v = torch.randn_like(theta), torch.zeros_like(theta), torch.zeros_like(theta)
    ops.adam_clip_step(theta[:N * K], grad[:N * K], m[:N * K], v[:N * K], 1e-2)   # the OTHER half
    assert not pw.stale
    ops.adam_clip_step(theta, grad, m, v, 1e-2)             # library-side write, no version bump
    assert pw.stale
    ops.repack_weights(W.device)
    assert not pw.stale
    assert run() <= TOL
    for _ in range(9):                                      # nobody asks for it any more
        ops.repack_weights(W.device)
    assert (W.data_ptr(), N, K, K, 1) not in ops._packs[ops._dev_key(W.device)]


def test_train_steps_with_and_without_x6_agree(hp, monkeypatch):
    '''three cfg-2-shaped Adam steps with the packed-weight products against the same steps on the
    exact-fp32 kernels: losses to 2e-5; 99.9 % of every parameter tensor to 1e-4 of its scale (two
    fp32-accurate evaluations of the same step; Adam's normalisation turns a rounding difference
    in a gradient that is itself ~0 into a full-size update, hence a quantile and not a max)'''
    from danet_amd import ops
    from test_gpu_fullsize import _setup, _synth
    res = []
    for on in (1, 0):
        monkeypatch.setattr(ops, 'GEMM_X6', on)
        model = _setup(hp, BATCH_SIZE=8, MAX_TRAIN_LEN=64)
        src = _synth(hp, 8, 64, 21)
        losses = [float(model.train_step(src)['loss']) for _ in range(3)]
        torch.cuda.synchronize()
        res.append((losses, {k: np.array(v) for k, v in model.param_dict().items()}))
    (l1, p1), (l0, p0) = res
    assert np.allclose(l1, l0, rtol=2e-5), (l1, l0)
    for k in p0:
        a, b = torch.as_tensor(p1[k]).double(), torch.as_tensor(p0[k]).double()
        close = (a - b).abs() <= 1e-4 * float(b.abs().max()) + 1e-7
        assert float(close.double().mean()) >= 0.999, (k, float(close.double().mean()))


# ---------------------------------------------------------------- TN: the weight gradients
TN_GROUPS = [
    # (K, [(M, N, lda, ldb, ldc, beta)]): a cfg-2 layer's dWx / dWh of both directions; the bottom
    # layer's (D = 129: ragged M, one K slice more); dWout; tiny / ragged everything
    (4096, [(600, 1200, 600, 1200, 1200, 0.0), (600, 1200, 600, 1200, 1200, 1.0),
            (300, 1200, 600, 1200, 1200, 0.0), (300, 1200, 600, 1200, 1200, 1.0)]),
    (1024, [(129, 1200, 132, 1200, 1200, 1.0), (300, 1200, 600, 1200, 1200, 0.0)]),
    (2048, [(600, 2580, 600, 2580, 2580, 0.0)]),
    (40, [(5, 7, 8, 8, 7, 1.0), (130, 129, 132, 132, 129, 0.0)]),
    (16, [(128, 128, 128, 128, 128, 0.0)]),
    # tail rows (1..4 rows beyond a multiple of 128 are fp32 FMA chains, not a tile row): without K
    # slices (straight into C, beta 1 and 0), three tail rows beside a plain product, 260 = 2 tiles + 4
    (48, [(131, 12, 132, 12, 16, 1.0), (129, 200, 132, 200, 200, 0.0), (64, 64, 64, 64, 64, 1.0)]),
    (4096, [(260, 600, 260, 600, 600, 1.0), (129, 1200, 132, 1200, 1200, 0.0)]),
    # K not a multiple of 16 far into the operands: the rows beyond K of the last k-tile must read as
    # zeros although the loads' scalar offset (the k-tile) is megabytes
    (4100, [(130, 64, 132, 64, 64, 0.0), (64, 200, 64, 200, 200, 1.0)]),
]


@pytest.mark.parametrize('K,shapes', TN_GROUPS)
def test_x6_tn_group_vs_float64(K, shapes):
    from danet_amd import ops
    g = torch.Generator(device='cuda').manual_seed(K + len(shapes))
    probs, refs, c0s = [], [], []
    for (M, N, lda, ldb, ldc, beta) in shapes:
        A = torch.randn(K, lda, device='cuda', generator=g) * torch.exp2(
            torch.randint(-5, 5, (K, lda), device='cuda', generator=g).float())
        Bm = torch.randn(K, ldb, device='cuda', generator=g)
        C = torch.randn(M, ldc, device='cuda', generator=g)
        c0s.append(C.clone())
        refs.append(A[:, :M].double().t() @ Bm[:, :N].double() + beta * C[:, :N].double())
        probs.append((A, lda, Bm, ldb, C, ldc, M, N, beta))
    assert ops._x6_tn_ok(probs, K)
    ops.gemm_group(probs, K, transA=True)
    outs = [pr[4].clone() for pr in probs]
    for (M, N, lda, ldb, ldc, beta), pr, ref, c0 in zip(shapes, probs, refs, c0s):
        C = pr[4]
        assert _err(C[:, :N], ref) <= TOL, (M, N, _err(C[:, :N], ref))
        if ldc > N:
            assert torch.equal(C[:, N:], c0[:, N:])          # nothing written beyond N
    # the exact-fp32 group on the same operands, and bit-reproducibility of the x6 one
    for pr, c0 in zip(probs, c0s):
        pr[4].copy_(c0)
    ops.gemm_group(probs, K, transA=True)
    for pr, o in zip(probs, outs):
        assert torch.equal(pr[4], o)


@pytest.mark.parametrize('M,N,lda,ldb', [(77, 131, 80, 132), (129, 1200, 132, 1200), (255, 50, 256, 52),
                                         (130, 66, 132, 68)])
def test_x6_tn_row_pads_never_reach_a_result(M, N, lda, ldb):
    '''the pad of every operand row (columns M..lda-1 / N..ldb-1) holds NaN: odd M / N share a split pair
    with the first pad value and take the masking kernel variant, even ones only feed accumulator rows
    that are never stored, 1..4 rows over a multiple of 128 are FMA chains that never read the pad
    (include/danet_hip.h: the pad need not be initialised)'''
    from danet_amd import ops
    K = 1030
    g = torch.Generator(device='cuda').manual_seed(M * N)
    A = torch.full((K, lda), float('nan'), device='cuda')
    Bm = torch.full((K, ldb), float('nan'), device='cuda')
    A[:, :M] = torch.randn(K, M, device='cuda', generator=g)
    Bm[:, :N] = torch.randn(K, N, device='cuda', generator=g)
    ldc = (N + 3) // 4 * 4
    C = torch.zeros(M, ldc, device='cuda')
    probs = [(A, lda, Bm, ldb, C, ldc, M, N, 0.0)]
    assert ops._x6_tn_ok(probs, K)
    ops.gemm_group(probs, K, transA=True)
    ref = A[:, :M].double().t() @ Bm[:, :N].double()
    assert torch.isfinite(C[:, :N]).all()
    assert _err(C[:, :N], ref) <= TOL
    assert torch.equal(C[:, N:], torch.zeros_like(C[:, N:]))


def test_x6_tn_transposition_and_identity():
    '''transpose-detecting: A = one-hot rows picks single rows of B exactly (the split is exact),
    and an asymmetric pattern lands at [m][n], not [n][m]'''
    from danet_amd import ops
    K, M, N = 64, 64, 96
    A = torch.zeros(K, M, device='cuda')
    A[torch.arange(M), torch.arange(M)] = 1.0                # A^T B = B[:M]
    g = torch.Generator(device='cuda').manual_seed(4)
    Bm = torch.randn(K, N, device='cuda', generator=g) * torch.exp2(
        torch.randint(-20, 20, (K, N), device='cuda', generator=g).float())
    C = torch.empty(M, N, device='cuda')
    ops.gemm_group([(A, M, Bm, N, C, N, M, N, 0.0)], K, transA=True)
    assert torch.equal(C, Bm[:M])



@pytest.mark.parametrize('M,N,K', [(4096, 2400, 1200), (1251, 1200, 600), (100, 36, 4100), (77, 130, 40)])
def test_x6_bias_and_k_major_weight(M, N, K):
    '''the hoisted input half of an LSTM layer: gates = x Wx + b with Wx stored [K][N]
    (stride_n = 1); sliced (few tiles, long K) and unsliced launches, vector and scalar epilogues'''
    from danet_amd import ops
    g = torch.Generator(device='cuda').manual_seed(M + N)
    A = torch.randn(M, K, device='cuda', generator=g)
    W = torch.randn(K, N, device='cuda', generator=g) * 0.05
    bias = torch.randn(N, device='cuda', generator=g)
    C = torch.empty(M, N, device='cuda')
    ops.gemm_w(A, K, W, 1, N, C, M, N, K, N, bias=bias)
    ref = A.double() @ W.double() + bias.double()
    assert _err(C, ref) <= TOL


def test_launch_events_time_a_recurrent_kernel_and_are_consumed(hp):
    '''danet_next_launch_events(start, stop): the pair rides on the next recurrent launch (what
    bench.py's roofline uses): the elapsed time is that kernel's, inside the host-side bracket; the
    slot is empty afterwards (the following launch does not touch the events again)'''
    from danet_amd import _lib, ops
    from test_gpu_fullsize import _setup, _synth
    model = _setup(hp, BATCH_SIZE=32, MAX_TRAIN_LEN=64)
    src = _synth(hp, 32, 64, 3)
    model.train_step(src)
    torch.cuda.synchronize()
    _lib.profile_start(only=('lstm_bwd',))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    model.train_step(src)
    e1.record()
    prof = _lib.profile_stop()
    n, ms = prof['lstm_bwd']
    assert n == hp.NUM_LSTM_LAYERS
    assert 0.05 * n < ms < e0.elapsed_time(e1)              # three BPTT kernels inside one step
    # nothing is left armed: an un-profiled step runs and the status stays clean
    model.train_step(src)
    torch.cuda.synchronize()
    assert ops.lstm_status_ok()
