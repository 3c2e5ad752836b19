'''
Round-3 GPU tests (run with -m gpu): ABI hygiene under host threads, the status word in
pinned host memory, the bounded run-ahead.
'''
import threading

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def cu(x, dtype=torch.float32):
    return torch.as_tensor(np.asarray(x)).to('cuda', dtype)


def relerr(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return np.abs(a - b).max() / (np.abs(b).max() + 1e-30)


def test_two_host_threads_launch_concurrently():
    '''Two host threads, each on its own HIP stream with its own workspace, launch stream-K
    grouped GEMMs (the launches that number themselves from the process-wide atomic counter and
    hand partial tiles over through flags in their workspace) and persistent LSTM layers at the
    same time; every result equals the single-threaded one bit for bit.'''
    from danet_amd import ops, _lib
    _lib.load()
    dev = torch.device('cuda')
    rng = np.random.RandomState(0)
    M, N, K = 300, 1200, 4096
    A = torch.as_tensor(rng.randn(K, M).astype(np.float32)).cuda()
    Bm = torch.as_tensor(rng.randn(K, N).astype(np.float32)).cuda()
    T, B, D, H = 24, 16, 40, 64
    x = torch.as_tensor(rng.randn(T * B, D).astype(np.float32)).cuda()
    Ws = [torch.as_tensor((rng.randn(D + H, 4 * H) * 0.1).astype(np.float32)).cuda() for _ in range(2)]
    bs = [torch.as_tensor((rng.randn(4 * H) * 0.1).astype(np.float32)).cuda() for _ in range(2)]

    def work(n_iter):
        outs = []
        for _ in range(n_iter):
            C = torch.empty(M, N, device=dev)
            ops.gemm_group([(A, M, Bm, N, C, N, M, N, 0.0)], K, transA=True, max_workgroups=256)
            c = ops.lstm_layer_fwd(x, D, D, T, B, H, Ws, bs)
            outs.append((C, c.ypad[1:T + 1].clone()))
        return outs

    ref = work(1)[0]
    torch.cuda.synchronize()
    results, errors = {}, []

    def thread_main(tid):
        try:
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):
                results[tid] = work(12)
            s.synchronize()
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=thread_main, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    torch.cuda.synchronize()
    assert not errors, errors
    assert ops.lstm_status_ok()
    for tid in (0, 1):
        for C, y in results[tid]:
            assert torch.equal(C, ref[0]) and torch.equal(y, ref[1])


def test_status_word_lives_in_pinned_host_memory_and_kernels_can_write_it(hp):
    '''single process: the persistent kernels' status word is pinned, device-mapped host
    memory (no per-step copy); a forced timeout is written by the GPU and read by the host
    directly, with the DANET_STATUS_TIMEOUT bit pattern (1.0f)'''
    from danet_amd import ops, _lib
    dev = torch.device('cuda')
    w = ops.status_word(dev)
    assert not w.is_cuda and w.is_pinned() and w.numel() == 4
    rng = np.random.RandomState(1)
    T, B, D, H = 8, 16, 16, 32
    x = torch.as_tensor(rng.randn(T * B, D).astype(np.float32)).cuda()
    Ws = [torch.as_tensor((rng.randn(D + H, 4 * H) * 0.1).astype(np.float32)).cuda() for _ in range(2)]
    bs = [torch.zeros(4 * H, device=dev) for _ in range(2)]
    ops.lstm_layer_fwd(x, D, D, T, B, H, Ws, bs)
    torch.cuda.synchronize()
    assert int(w[0]) == 0
    _lib.set_option('lstm_fault_inject', 1)
    _lib.set_option('lstm_spin_limit', 2048)
    ops.lstm_layer_fwd(x, D, D, T, B, H, Ws, bs)
    torch.cuda.synchronize()
    assert int(w[0]) == 0x3F800000
    assert w.view(torch.float32)[0].item() == 1.0
    assert not ops.lstm_status_ok() and int(w[0]) == 0


@pytest.mark.parametrize('act', [0, 1])
@pytest.mark.parametrize('mode', [0, 1])
@pytest.mark.parametrize('B,C,T,F,E', [(3, 2, 9, 33, 20), (2, 3, 17, 129, 40), (1, 1, 5, 7, 4),
                                       (2, 2, 40, 129, 20), (2, 2, 11, 33, 16), (2, 3, 9, 17, 7),
                                       (1, 2, 300, 129, 33)])   # E != EP: the guarded row accesses
def test_fused_separator_pit_matches_unfused_and_oracle(act, mode, B, C, T, F, E):
    '''danet_separate_pit_fwd / _bwd (separator + phase re-attach + PIT-MSE + SNR in one pass,
    app/modules.py:548-603 -> main.py:281-290 -> app/ops.py:374-431) against (i) the two-kernel
    HIP path -- same arithmetic per bin, so loss / SNR agree to reduction-order rounding and
    the permutation index and dembed / dattr match tightly -- and (ii) the float64 oracle'''
    from danet_amd import ops
    from oracle import danet_oracle as O
    rng = np.random.RandomState(B * 1000 + C * 100 + T + E + act * 7 + mode)
    N = T * F
    embed = (rng.randn(B, N, E) * 0.7).astype(np.float32)
    attr = (rng.randn(B, C, E) * 0.8).astype(np.float32)
    src = ((rng.randn(B, C, T, F) + 1j * rng.randn(B, C, T, F)) * 4).astype(np.complex64)
    src[:, :, 0] = 0                                       # an all-zero frame (padded batches)
    fe = O.frontend(src)
    cu = lambda a: torch.as_tensor(np.ascontiguousarray(a)).cuda()
    mix_pwr = cu(fe['mix_pwr'].astype(np.float32))
    ph = np.stack([np.cos(fe['phase']), np.sin(fe['phase'])], -1).astype(np.float32)
    phasor = cu(ph)
    s_src = cu(src)

    def run(fused):
        e = cu(embed).requires_grad_(True)
        a = cu(attr).requires_grad_(True)
        if fused:
            loss, _, idx, snr = ops.separate_pit_loss(mix_pwr, a, e, s_src, phasor, act, mode=mode, eps=1e-7)
        else:
            sep, _ = ops.SeparateFn.apply(mix_pwr, a, e, act, False)
            loss, _, idx, snr = ops.pit_mse_loss(s_src, sep, phasor, mode=mode, eps=1e-7)
        loss.backward()
        torch.cuda.synchronize()
        return (float(loss), float(snr), idx.cpu().numpy(), e.grad.cpu().numpy(), a.grad.cpu().numpy())

    lf, sf, pf, def_, daf = run(True)
    lu, su, pu, deu, dau = run(False)
    assert np.array_equal(pf, pu)
    assert abs(lf - lu) <= 2e-6 * abs(lu) and abs(sf - su) <= 1e-5 * max(abs(su), 1.0), (lf, lu, sf, su)
    # (C = 1: the softmax mask is identically 1 and both gradients are exact zeros up to
    # cancellation noise -- compare on the scale of the incoming gradient instead)
    gscale = float(np.abs(src).max()) / (B * N)
    assert np.abs(def_ - deu).max() <= 1e-6 * np.abs(deu).max() + 1e-6 * gscale, \
        (np.abs(def_ - deu).max(), np.abs(deu).max())
    assert np.abs(daf - dau).max() <= 2e-5 * np.abs(dau).max() + 1e-5 * gscale * N ** 0.5, \
        (np.abs(daf - dau).max(), np.abs(dau).max())
    # oracle (float64): loss and permutation
    sep64, _ = O.sep_dot(fe['mix_pwr'], attr.astype(np.float64), embed.astype(np.float64),
                         'softmax' if act == 0 else 'sigmoid', return_masks=True)
    if mode == 0:
        est = O.reattach_phase(sep64, fe['phase'])
        lo, _, io, _ = O.pit_mse_loss(src.astype(np.complex128), est)
    else:
        lo, _, io, _ = O.pit_mse_loss(fe['src_pwr'], sep64)
    # (C = 1, mode 0: estimate == mixture == the one source, the loss is rounding noise)
    assert abs(lf - lo) <= 1e-4 * abs(lo) + 1e-9 * float(np.mean(np.abs(src) ** 2))
    assert np.array_equal(pf, io)


def test_train_step_uses_fused_heads_and_matches_unfused(hp):
    '''Model.train_step takes the fused separator + loss kernels; three steps give the same
    parameters as the unfused path to rounding'''
    from danet_amd.model import Model
    res = []
    for fuse in (True, False):
        hp.reset()
        hp.load(dict(BATCH_SIZE=3, MAX_N_SIGNAL=2, FFT_SIZE=32, FFT_STRIDE=8, EMBED_SIZE=8,
                     NUM_LSTM_LAYERS=2, LSTM_HDIM=16, NUM_ANCHOR=4, ENCODER_TYPE='bilstm-orig',
                     TRAIN_ESTIMATOR_METHOD='anchor', INFER_ESTIMATOR_METHOD='anchor',
                     SEPARATOR_TYPE='dot-softmax-orig'))
        hp.digest()
        model = Model('fh', device='cuda', seed=11).build()
        model.fuse_heads = fuse
        rng = np.random.RandomState(3)
        src = torch.as_tensor(((rng.randn(3, 2, 10, 17) + 1j * rng.randn(3, 2, 10, 17)) * 5)
                              .astype(np.complex64)).cuda()
        out = None
        for _ in range(3):
            out = model.train_step(src)
        res.append((float(out['loss']), model.param_dict()))
        if fuse:
            o = model.forward(src, fuse_heads=True)
            assert 'sep_pwr' not in o and 'loss' in o
    assert abs(res[0][0] - res[1][0]) <= 1e-5 * abs(res[1][0])
    for k in res[0][1]:
        a, b = res[0][1][k], res[1][1][k]
        assert np.abs(a - b).max() <= 1e-4 * (np.abs(b).max() + 1e-12), k


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


@pytest.mark.parametrize('E', [20, 6])      # 6: E != EP, the guarded row accesses
@pytest.mark.parametrize('est,sepn,C', [('anchor', 'dot-softmax-orig', 2), ('truth-weighted', 'dot-softmax-orig', 3),
                                        ('truth', 'dot-sigmoid-orig', 2), ('truth-threshold', 'dot-softmax-orig', 2)])
def test_estimator_backward_recomputes_the_separator_term(hp, monkeypatch, est, sepn, C, E):
    '''inside train_step with the anchor estimator the fused separator + loss backward only
    produces dattr and danet_attractor_anchor_bwd_embed_sep forms the whole embedding gradient
    in one pass (the separator's term is not written to HBM and read back): same additions in the
    same order -> the parameters after three steps equal the two-pass form's'''
    from danet_amd.model import Model
    from danet_amd import ops
    res = []
    for recompute in (1, 0):
        monkeypatch.setattr(ops, 'HEADS_RECOMPUTE', recompute)
        hp.reset()
        hp.load(dict(BATCH_SIZE=4, MAX_N_SIGNAL=C, FFT_SIZE=64, FFT_STRIDE=16, EMBED_SIZE=E,
                     NUM_LSTM_LAYERS=2, LSTM_HDIM=16, NUM_ANCHOR=6, ENCODER_TYPE='bilstm-orig',
                     TRAIN_ESTIMATOR_METHOD=est, INFER_ESTIMATOR_METHOD='anchor',
                     SEPARATOR_TYPE=sepn))
        hp.digest()
        model = Model('rc', device='cuda', seed=3).build()
        model.keep_grads = True
        rng = np.random.RandomState(5)
        src = torch.as_tensor(((rng.randn(4, C, 70, 33) + 1j * rng.randn(4, C, 70, 33)) * 5)
                              .astype(np.complex64)).cuda()
        for _ in range(3):
            out = model.train_step(src)
        torch.cuda.synchronize()
        res.append((float(out['loss']), model.param_dict(), model.grad_dict()))
    assert res[0][0] == res[1][0]
    for k in res[0][1]:
        a, b = res[0][2][k], res[1][2][k]
        assert np.abs(a - b).max() <= 1e-6 * (np.abs(b).max() + 1e-30), ('grad', k)
        a, b = res[0][1][k], res[1][1][k]
        assert np.abs(a - b).max() <= 1e-6 * (np.abs(b).max() + 1e-30), ('param', k)


def test_deferred_bias_gradient_reduce_is_bit_identical(hp, monkeypatch):
    '''danet_lstm_bwd(db, DANET_LSTM_DB_DEFERRED) + danet_lstm_bwd_db_reduce on the side chain: the
    same partials summed by the same kernel, only later -> gradients and parameters bit-equal'''
    from danet_amd.model import Model
    from danet_amd import ops
    res = []
    for defer in (True, False):
        monkeypatch.setattr(ops, 'DB_DEFER', defer)
        hp.reset()
        hp.load(dict(BATCH_SIZE=32, MAX_N_SIGNAL=2, FFT_SIZE=64, FFT_STRIDE=16, EMBED_SIZE=20,
                     NUM_LSTM_LAYERS=3, LSTM_HDIM=64, NUM_ANCHOR=6, ENCODER_TYPE='bilstm-orig',
                     TRAIN_ESTIMATOR_METHOD='anchor', INFER_ESTIMATOR_METHOD='anchor',
                     SEPARATOR_TYPE='dot-softmax-orig'))
        hp.digest()
        model = Model('dbd', device='cuda', seed=3).build()
        model.keep_grads = True
        rng = np.random.RandomState(5)
        src = torch.as_tensor(((rng.randn(32, 2, 40, 33) + 1j * rng.randn(32, 2, 40, 33)) * 5)
                              .astype(np.complex64)).cuda()
        for _ in range(3):
            out = model.train_step(src)
        torch.cuda.synchronize()
        assert ops.lstm_status_ok()
        res.append((float(out['loss']), model.param_dict(), model.grad_dict()))
    assert res[0][0] == res[1][0]
    nb = 0
    for k in res[0][1]:
        if k.endswith('/B'):
            nb += 1
            assert np.array_equal(res[0][2][k], res[1][2][k]), ('grad', k)
            assert np.abs(res[0][2][k]).max() > 0
    assert nb >= 3


def test_lstm_bwd_db_reduce_entry_point():
    '''the C entry points directly: deferred launch leaves db untouched, the reduce call then
    produces exactly what the undeferred launch writes'''
    from danet_amd import _lib
    L = _lib.load()
    T, B, H = 12, 32, 64
    if L.danet_lstm_bwd_db_supported(T, B, H, 2) != 1:
        pytest.skip('outside the reduce-scatter geometry')
    g = torch.Generator(device='cuda').manual_seed(1)
    rnd = lambda *s: torch.randn(*s, device='cuda', generator=g)
    dy = rnd(T, B, 2 * H)
    Wh = [rnd(H, 4 * H) * 0.1 for _ in range(2)]
    gates = [torch.sigmoid(rnd(T * B, 4 * H)) for _ in range(2)]
    cells = [rnd((T + 1) * B, H).tanh() for _ in range(2)]
    wn = _lib.ws_bytes(_lib.WS_LSTM, T, B, H, 2)
    outs = []
    for flags in (0, 2):
        ws = torch.zeros(wn, dtype=torch.uint8, device='cuda')
        st = torch.zeros(1, dtype=torch.int32, device='cuda')
        da = [torch.empty(T * B, 4 * H, device='cuda') for _ in range(2)]
        db = [torch.full((4 * H,), 7.0, device='cuda') for _ in range(2)]
        p = _lib.ptr
        rc = L.danet_lstm_bwd(_lib.stream(), T, B, H, 2, p(dy), 2 * H, p(Wh[0]), p(Wh[1]), 4 * H,
                              p(gates[0]), p(gates[1]), p(cells[0]), p(cells[1]), p(da[0]), p(da[1]),
                              p(db[0]), p(db[1]), 0.0, p(ws), wn, p(st), flags)
        assert rc == 0, L.danet_last_error()
        if flags:
            torch.cuda.synchronize()
            assert all(bool((t == 7.0).all()) for t in db)
            rc = L.danet_lstm_bwd_db_reduce(_lib.stream(), T, B, H, 2, p(db[0]), p(db[1]), 0.0, p(ws), wn)
            assert rc == 0, L.danet_last_error()
        torch.cuda.synchronize()
        assert int(st) == 0
        outs.append([t.cpu().numpy() for t in db + da])
    for a, b in zip(*outs):
        assert np.array_equal(a, b)
    ref = outs[0][2].sum(axis=0)
    assert np.abs(outs[0][0] - ref).max() <= 1e-4 * np.abs(ref).max()


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
