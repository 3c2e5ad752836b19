'''
CPU tests (run with -m "not gpu"): the oracle against the reference-captured
golden vectors (G1-G4), the known-answer tests K1-K10 on the oracle itself,
float64 finite-difference gradient checks of the torch restatement (K11), and
agreement of the two independent restatements.
'''
import itertools
import os
import random

import numpy as np
import pytest
import torch

from oracle import danet_oracle as O
from oracle import torch_ref as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'frontend_ref.npz')


@pytest.fixture(scope='module')
def gold():
    return np.load(GOLD)


# ------------------------------------------------------------- G1-G4 (pinned)
def test_window_bit_exact(gold):
    assert np.array_equal(O.fft_window(256).view(np.uint32), gold['wnd256_bits'])
    assert np.array_equal(O.fft_window(512).view(np.uint32), gold['wnd512_bits'])
    w = O.fft_window(256)
    assert w[0] == 0.0 and w[-1] == 0.0          # symmetric Hann


@pytest.mark.parametrize('L', [256, 257, 319, 320, 8000, 8001, 8063, 8064])
def test_stft_matches_reference(gold, L):
    x = np.random.RandomState(0).randn(L).astype(np.float32)
    w = O.fft_window(256)
    X = O.stft(x, w, 256, 64)
    ref = gold['stft256_L%d' % L]
    assert X.shape == ref.shape and X.dtype == ref.dtype == np.complex64
    assert X.shape[0] == 1 + -(-L // 64) == O.stft_frame_count(L, 256, 64)     # K10
    assert np.abs(X - ref).max() <= 2e-7 * np.abs(ref).max()


def test_stft_int16_scale_and_short_input(gold):
    w = O.fft_window(256)
    X = O.stft(gold['stft256_int16scale_x'], w, 256, 64)
    assert np.abs(X - gold['stft256_int16scale']).max() <= 2e-7 * np.abs(gold['stft256_int16scale']).max()
    assert int(gold['stft256_short_raises']) == 1
    with pytest.raises(ValueError):
        O.stft(np.zeros(255, np.float32), w, 256, 64)


def test_stft_512_long(gold):
    x = np.random.RandomState(0).randn(160000).astype(np.float32)
    X = O.stft(x, O.fft_window(512), 512, 128)
    assert X.shape == tuple(gold['stft512_L160000_shape']) == (1251, 257)
    sc = np.abs(X).max()
    assert np.abs(X[:2] - gold['stft512_L160000_head']).max() < 2e-7 * sc
    assert np.abs(X[-2:] - gold['stft512_L160000_tail']).max() < 2e-7 * sc
    assert np.abs(X[600:602] - gold['stft512_L160000_mid']).max() < 2e-7 * sc


@pytest.mark.parametrize('L', [256, 257, 320, 8000, 8064])
def test_istft_matches_reference(gold, L):
    w = O.fft_window(256)
    y = O.istft(gold['stft256_L%d' % L], 64, w)
    ref = gold['istft256_L%d' % L]
    assert y.shape == ref.shape
    assert np.abs(y - ref).max() <= 1e-12 * max(1.0, np.abs(ref).max())
    T = gold['stft256_L%d' % L].shape[0]
    assert O.istft_num_frames_used(T, 256, 64) == max(T - 256 // 64, 0)   # drops N/S frames


def test_istft_does_not_undo_scaling():
    w = O.fft_window(256)
    x = np.random.RandomState(3).randn(4000).astype(np.float32)
    y = O.istft(O.stft(x, w, 256, 64), 64, w)
    wsum = float(w.astype(np.float64).sum())
    assert abs(wsum - 162.336) < 1e-2
    assert np.abs(y[256:3500] * wsum - x[128:3372]).max() < 1e-4 * np.abs(x).max()


def test_random_zeropad_matches_reference(gold):
    base = gold['zeropad_base']
    for k in range(4):
        random.seed(k)
        assert np.array_equal(O.random_zeropad(base, 5, axis=0), gold['zeropad_seed%d_axis0' % k])
        random.seed(k)
        assert np.array_equal(O.random_zeropad(base, 7, axis=-1), gold['zeropad_seed%d_axis-1' % k])


# --------------------------------------------------------------- KATs (K1-K9)
def test_k1_k2_lstm_gate_order_no_tanh():
    H, D, B = 3, 2, 2
    W = np.zeros((D + H, 4 * H))
    b = np.concatenate([np.full(H, 1.0), np.full(H, 1.5), np.full(H, -1.0), np.full(H, 1.0)])
    x = np.random.RandomState(0).randn(B, 2, D)
    out, cells = O.lyr_lstm(x, W, b, H, return_cell=True)
    assert np.allclose(cells[:, 0], 0.817574, atol=1e-6)
    assert np.allclose(out[:, 0], 0.492549, atol=1e-6)
    assert np.allclose(cells[:, 1], 1.037454, atol=1e-6)
    assert np.all(O.lyr_lstm(x, W, O.lstm_bias_init(H), H) == 0.0)            # K2


def test_k3_bilstm_time_reversal():
    rng = np.random.RandomState(1)
    B, T, D, H = 2, 6, 3, 4
    x = rng.randn(B, T, D)
    Wf, Wb = rng.randn(D + H, 4 * H) * 0.3, rng.randn(D + H, 4 * H) * 0.3
    bf, bb = rng.randn(4 * H) * 0.1, rng.randn(4 * H) * 0.1
    y = O.lyr_bilstm(x, Wf, bf, Wb, bb, H)
    rev = O.lyr_lstm(x[:, ::-1], Wb, bb, H)
    for t in range(T):
        assert np.allclose(y[:, t, H:], rev[:, T - 1 - t])
    assert np.allclose(y[..., :H], O.lyr_lstm(x, Wf, bf, H))


def test_k4_equal_attractors():
    rng = np.random.RandomState(2)
    B, C, T, F, E = 1, 3, 2, 4, 5
    embed = rng.randn(B, T * F, E)
    attr = np.repeat(rng.randn(B, 1, E), C, axis=1)
    mix = rng.rand(B, T, F) + 1
    sep = O.sep_softmax(mix, attr, embed)
    assert np.allclose(sep, np.repeat((mix / C)[:, None], C, axis=1))
    sep = O.sep_sigmoid(mix, np.zeros((B, C, E)), embed)
    assert np.allclose(sep, np.repeat((mix * 0.5)[:, None], C, axis=1))


def test_k5_k6_truth_estimators():
    embed = np.array([2.0, 4.0, 6.0]).reshape(1, 1, 1, 3)
    src_pwr = np.array([1.0, 0.5]).reshape(1, 2, 1, 1)
    a = O.est_truth(embed, src_pwr)
    assert np.allclose(a[0, 0], [1, 2, 3]) and np.all(a[0, 1] == 0)            # count + 1
    five = np.float32(5.0)
    mix = np.array([five, np.nextafter(five, np.float32(10))], dtype=np.float32).reshape(1, 1, 2)
    emb = np.eye(2, 3, dtype=np.float32).reshape(1, 1, 2, 3)
    sp = np.ones((1, 2, 1, 2), np.float32); sp[0, 1] = 0.5
    a = O.est_truth_threshold(emb, sp, mix)
    assert abs(a[0, 0, 0]) < 1e-6 and abs(a[0, 0, 1] - 1.0) < 1e-5             # 5.0 excluded


def test_k7_anchor_diagonal_participates():
    rng = np.random.RandomState(3)
    embed = rng.randn(1, 6, 9, 3) * np.array([3.0, 0.2, 0.2])
    anchors = np.array([[4.0, 0, 0], [-4.0, 0, 0], [0, 1.0, 0], [0, 0, 1.0]])
    _, info = O.est_anchor(embed, anchors, 2, return_all=True)
    gram = info['asets'] @ np.swapaxes(info['asets'], -1, -2)
    full_max = gram.max(axis=(-1, -2))
    assert int(np.argmin(full_max[0])) == int(info['subset_choice'][0])
    offdiag = np.array([g[0, 1] for g in gram[0]])
    assert int(np.argmin(offdiag)) != int(info['subset_choice'][0])
    assert np.array_equal(O.combinations(4, 2), np.array(list(itertools.combinations(range(4), 2))))


def test_k8_pit_swap_and_perm_order():
    rng = np.random.RandomState(4)
    x = rng.rand(1, 2, 3, 5) + np.array([0.0, 10.0]).reshape(1, 2, 1, 1)
    l0, perms, i0, _ = O.pit_mse_loss(x, x * 1.01)
    l1, _, i1, _ = O.pit_mse_loss(x, (x * 1.01)[:, ::-1])
    assert int(i0[0]) == 0 and int(i1[0]) == 1 and abs(l0 - l1) < 1e-12
    x3 = rng.rand(1, 3, 2, 2) + np.array([0.0, 10.0, 100.0]).reshape(1, 3, 1, 1)
    for p, perm in enumerate(itertools.permutations(range(3))):
        y = np.zeros_like(x3)
        for i, j in enumerate(perm):
            y[:, j] = x3[:, i]
        assert int(O.pit_mse_loss(x3, y)[2][0]) == p
    # sums over speakers, means over T*F, means over batch
    xx, yy = rng.rand(2, 2, 3, 4), rng.rand(2, 2, 3, 4)
    loss, perms, idx, sets = O.pit_mse_loss(xx, yy)
    manual = np.mean([min(sum(((xx[b, i] - yy[b, pm[i]]) ** 2).mean() for i in range(2))
                          for pm in itertools.permutations(range(2))) for b in range(2)])
    assert abs(loss - manual) < 1e-12


def test_k9_argmax_ties_lowest_index():
    embed = np.ones((1, 1, 2, 2))
    src_pwr = np.zeros((1, 2, 1, 2))               # all-zero frame: tie between speakers
    mix = np.ones((1, 1, 2))
    a = O.est_truth_weighted(embed, src_pwr, mix)
    assert np.allclose(a[0, 0], 1.0, atol=1e-6) and np.all(a[0, 1] == 0.0)
    sim = np.array([[3.0, 1.0, 1.0]])
    assert int(np.argmin(sim, axis=1)[0]) == 1


def test_tf_adam_eps_outside_root():
    th, m, v = O.tf_adam_step(np.array([1.0]), np.array([0.5]), np.zeros(1), np.zeros(1),
                              t=1, lr=0.1, clip=None)
    # t=1: m=0.05, v=2.5e-4; lr_t = 0.1*sqrt(1-.999)/(1-.9); update = lr_t*m/(sqrt(v)+eps)
    lr_t = 0.1 * np.sqrt(1 - 0.999) / (1 - 0.9)
    assert abs(th[0] - (1.0 - lr_t * 0.05 / (np.sqrt(2.5e-4) + 1e-8))) < 1e-15
    th2, _, _ = O.tf_adam_step(np.array([1.0]), np.array([500.0]), np.zeros(1), np.zeros(1),
                               t=1, lr=0.1, clip=100.0)
    th3, _, _ = O.tf_adam_step(np.array([1.0]), np.array([100.0]), np.zeros(1), np.zeros(1),
                               t=1, lr=0.1, clip=None)
    assert th2[0] == th3[0]                                                    # value clip


# ----------------------------------------- two restatements agree + K11 (FD)
CFGS = [('truth-weighted', 'dot-sigmoid-orig'), ('anchor', 'dot-softmax-orig'),
        ('truth', 'dot-softmax-orig'), ('truth-threshold', 'dot-sigmoid-orig')]


def _tiny(seed=0, C=2):
    rng = np.random.RandomState(seed)
    B, T, F, E, H, L, A = 2, 8, 5, 3, 4, 2, 4
    src = (rng.randn(B, C, T, F) + 1j * rng.randn(B, C, T, F)) * 3
    p = O.init_bilstm_params(rng, F, E, H, L)
    p['global/train_estimator/anchors'] = rng.randn(A, E)
    p['global/infer_estimator/anchors'] = rng.randn(A, E)
    return src, p, dict(H=H, L=L, E=E, C=C, A=A)


@pytest.mark.parametrize('train_est,sepn', CFGS)
@pytest.mark.parametrize('C', [2, 3])
def test_numpy_and_torch_restatements_agree(train_est, sepn, C):
    src, p, dims = _tiny(1, C)
    cfg = dict(dims, train_est=train_est, infer_est='anchor', separator=sepn, with_valid=True)
    o = O.model_forward(src, p, cfg)
    r = R.model_forward(torch.tensor(src), {k: torch.tensor(v) for k, v in p.items()}, cfg)
    for k in ('embed', 'attrs', 'sep_pwr', 'loss', 'SNR', 'valid_loss', 'sep_pwr_valid'):
        a, b = np.asarray(o[k]), r[k].numpy()
        assert np.abs(a - b).max() <= 1e-12 * (np.abs(a).max() + 1e-30), k
    assert np.array_equal(o['perm_idx'], r['perm_idx'].numpy())


def test_float32_oracle_within_1e4_of_float64():
    src, p, dims = _tiny(2)
    cfg = dict(dims, train_est='anchor', infer_est='anchor', separator='dot-softmax-orig')
    o64 = O.model_forward(src, p, cfg, dtype=np.float64)
    o32 = O.model_forward(src.astype(np.complex64), p, cfg, dtype=np.float32)
    for k in ('embed', 'attrs', 'sep_pwr'):
        assert np.abs(o32[k] - o64[k]).max() <= 1e-4 * np.abs(o64[k]).max(), k


@pytest.mark.parametrize('train_est,sepn', CFGS[:2])
def test_k11_finite_difference_gradients(train_est, sepn):
    src, p, dims = _tiny(3)
    cfg = dict(dims, train_est=train_est, infer_est='anchor', separator=sepn)
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in p.items()}
    R.model_forward(torch.tensor(src), tp, cfg)['loss'].backward()
    rng = np.random.RandomState(0)
    eps = 1e-6
    for k in sorted(p):
        if tp[k].grad is None:
            continue
        for _ in range(3):
            idx = tuple(rng.randint(0, s) for s in p[k].shape)
            pp = {n: v.copy() for n, v in p.items()}
            pp[k][idx] += eps
            lp = O.model_forward(src, pp, cfg)['loss']
            pp[k][idx] -= 2 * eps
            lm = O.model_forward(src, pp, cfg)['loss']
            fd = (lp - lm) / (2 * eps)
            g = float(tp[k].grad[idx])
            # central-difference roundoff floor: ~1e-16 * |loss| / eps
            assert abs(fd - g) <= 1e-5 * max(abs(g), abs(fd)) + 1e-7, (k, idx, fd, g)


def test_lstm_orig_restatements_agree():
    rng = np.random.RandomState(5)
    B, T, F, E, H, L = 2, 5, 4, 3, 4, 2
    x = rng.randn(B, T, F)
    p = O.init_bilstm_params(rng, F, E, H, L, bidirectional=False)
    a = O.lstm_encoder(x, p, H, L, E)
    b = R.lstm_encoder(torch.tensor(x), {k: torch.tensor(v) for k, v in p.items()}, H, L, E).numpy()
    assert np.abs(a - b).max() < 1e-12


# ----------------------------------------------------- G5: the oracle is frozen
@pytest.mark.parametrize('name', sorted(__import__('oracle.g5', fromlist=['CASES']).CASES))
def test_g5_live_oracle_matches_committed_fixtures(name):
    '''SURVEY 8c G5 (debug_fetches keys main.py:389-397, app/modules.py:540-543,571,600):
    the LIVE oracle must reproduce the committed oracle-generated fixtures.  Integer
    tensors exactly; float64 tensors to 1e-11 of the tensor maximum -- the result of a
    float64 matmul may differ in the last bits between BLAS builds / CPU types, while any
    edit of the restated arithmetic moves results by many orders more.  An intended oracle
    change regenerates the fixtures with tests/golden/make_oracle_g5.py.'''
    from oracle import g5
    fix = dict(np.load(os.path.join(os.path.dirname(__file__), 'golden', 'oracle_g5_%s.npz' % name)))
    assert 'NOT produced by the reference' in str(fix.pop('_label'))
    live = g5.case_outputs(name)
    errs = g5.compare(fix, live, 1e-11)
    need = {'embed', 'attrs', 'masks', 'asets', 'subset_choice', 'output', 'loss', 'SNR',
            'perm_idx'}
    have = {k.split('__')[0] for k in errs}
    assert need <= have, need - have
    assert any(k.startswith('grad:') for k in errs)
    bad = {k: e for k, e in errs.items() if not e <= 1e-11}
    assert not bad, bad


# ------------------------------------------------------------------ round 4: inference fetches
def test_infer_forward_agrees_with_the_train_graph_and_numpy_oracle():
    '''oracle/torch_ref.infer_forward (main.py:384-385, :333-335, :685-690) == the pieces of the
    training graph evaluated on the mixture alone, and == the numpy oracle's functions (two
    restatements must agree): embedding, anchor attractors, masks, separated spectra'''
    import torch
    from oracle import torch_ref as R
    rng = np.random.RandomState(3)
    B, C, T, F, E, H, L, A = 2, 2, 9, 5, 3, 4, 2, 4
    params = O.init_bilstm_params(rng, F, E, H=H, L=L)
    params['global/infer_estimator/anchors'] = rng.randn(A, E)
    src = rng.randn(B, C, T, F) + 1j * rng.randn(B, C, T, F)
    mix = src.sum(axis=1)
    cfg = dict(H=H, L=L, E=E, C=C, A=A, infer_est='anchor', separator='dot-softmax-orig')
    tp = {k: torch.tensor(v, dtype=torch.float64) for k, v in params.items()}
    r = R.infer_forward(torch.tensor(mix), tp, cfg)
    fe = O.frontend(mix[:, None])
    emb = O.bilstm_encoder(fe['mix_log'], params, H, L, E)
    attr = O.est_anchor(emb, params['global/infer_estimator/anchors'], C)
    sep_pwr, masks = O.sep_dot(fe['mix_pwr'], attr, emb.reshape(B, -1, E), 'softmax', return_masks=True)
    sep = O.reattach_phase(sep_pwr, fe['phase'])
    for got, want in ((r['embed'], emb), (r['attrs'], attr), (r['masks'], masks), (r['sep_pwr'], sep_pwr),
                      (r['sep'], sep)):
        assert np.abs(got.numpy() - want).max() <= 1e-11 * (np.abs(want).max() + 1e-30)
    # softmax masks: the separated spectra add up to the mixture
    assert np.abs(r['sep'].sum(dim=1).numpy() - mix).max() < 1e-12 * np.abs(mix).max()


def test_est_kmeans_restatement():
    '''oracle/torch_ref.est_kmeans (extension; README.md:216 of the reference has no k-means):
    0 iterations == the anchor estimator; on two well separated clusters the iterations end at the
    |mix|-weighted cluster means (app/modules.py:476-482 with estimated assignments); a fixed
    point stays fixed'''
    import torch
    from oracle import torch_ref as R
    rng = np.random.RandomState(5)
    T, F, E, A, C = 40, 9, 6, 5, 2
    centres = rng.randn(C, E) * 3.0
    assign = rng.randint(0, C, size=(1, T, F))
    emb = torch.tensor(centres[assign] + 0.05 * rng.randn(1, T, F, E))
    w = torch.tensor(np.abs(rng.randn(1, T, F)) + 0.1)
    anchors = torch.tensor(rng.randn(A, E))
    a0 = R.est_kmeans(emb, anchors, C, w, iters=0)
    assert torch.equal(a0, R.est_anchor(emb, anchors, C))
    a10 = R.est_kmeans(emb, anchors, C, w, iters=10)
    a11 = R.est_kmeans(emb, anchors, C, w, iters=11)
    assert float((a10 - a11).abs().max()) < 1e-12              # converged: a fixed point
    ef, wf, lab = emb.reshape(-1, E).numpy(), w.reshape(-1).numpy(), assign.reshape(-1)
    truth = np.stack([(ef[lab == c] * wf[lab == c][:, None]).sum(0) / (wf[lab == c].sum() + 1e-7)
                      for c in range(C)])
    got = a10[0].numpy()
    # (the start attractors may label the clusters in either order)
    err = min(np.abs(got - truth).max(), np.abs(got - truth[::-1]).max())
    start = R.est_anchor(emb, anchors, C)[0].numpy()
    both_found = len(set(np.argmax(truth @ start.T, axis=1))) == 2
    assert (not both_found) or err < 1e-6 * np.abs(truth).max()


def test_parity_report_logits_entry():
    '''oracle/parity.py: the derived `logits` entry (embed . attr^T, app/modules.py:587-589, relative
    to the largest |logit|) exists, is exact for the float64 oracle's own tensors, reads a
    perturbed embedding as a failure and leaves the other entries alone'''
    from oracle import g5, parity as P
    src, params, cfg = g5.case_inputs('tiny_anchor_softmax')
    r64, _ = P.oracle_pair(src, params, cfg)
    got = {k: np.asarray(r64[k]) for k in P.KEYS}
    got['perm_idx'] = r64['perm_idx']
    rep = P.parity_report(got, src, params, cfg)
    assert rep['ok'] and rep['logits']['ok']
    assert rep['logits']['hip_vs_f64']['max_rel'] == 0.0
    assert rep['logits']['bound'] >= P.BAR_LOGITS and rep['logits']['max_abs_logit'] > 0
    bad = dict(got, embed=got['embed'] * (1 + 1e-3))
    rep_b = P.parity_report(bad, src, params, cfg)
    assert not rep_b['logits']['ok'] and not rep_b['ok']
    assert rep_b['logits']['hip_vs_f64']['max_rel'] > 1e-4
