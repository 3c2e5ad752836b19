'''
CPU ORACLE -- TEST INFRASTRUCTURE ONLY.

Literal numpy restatement of the reference's Deep-Attractor-Network hot path
(khaotik/DaNet-Tensorflow).  Every function cites the reference file:line it
follows.  Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s
`cpu_baseline` leg may import this; the product package
(`danet-tensorflow_amd/`) never does and fails loudly without its HIP library.

PARITY PINNING
  * STFT / iSTFT / random_zeropad (a1, a2, G1-G4): PINNED -- checked against
    outputs of the reference's own `app/utils.py` + `scipy.signal.stft`
    captured in the build container (`tests/golden/make_golden.py` ->
    `tests/golden/frontend_*.npz`).
  * Model path (a3-a16: front-end, BiLSTM, estimators, separators, PIT loss,
    SNR): PARITY UNPINNED by the reference -- it ships no tests / golden
    vectors, and TensorFlow 1.x (its runtime) cannot be installed here.  These
    functions are pinned instead by (i) known-answer tests K1-K11
    (tests/test_oracle_kat.py), (ii) an independent torch-CPU autograd
    restatement (oracle/torch_ref.py) that must agree in float64, and
    (iii) float64 finite differences for every gradient.

`dtype` selects the arithmetic type: float64 for the parity authority, float32
for a type-faithful rerun of the reference's FLOATX (`default.json:2`).
'''
import itertools
import math
import random

import numpy as np


# --------------------------------------------------------------------------
# a1 / G1: window + STFT      (reference: default.json:7, app/utils.py:117-122)
# --------------------------------------------------------------------------
def hann_sym(n):
    '''scipy.signal.hann(n) == scipy.signal.windows.hann(n, sym=True):
    w[k] = 0.5 - 0.5 cos(2 pi k / (n-1)); endpoints are exactly 0.'''
    if n == 1:
        return np.ones(1)
    k = np.arange(n, dtype=np.float64)
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * k / (n - 1))


def fft_window(n, floatx='float32'):
    '''default.json:7  np.sqrt(scipy.signal.hann(N)).astype(FLOATX)'''
    return np.sqrt(hann_sym(n)).astype(floatx)


def stft_frame_count(n_samples, nperseg, stride):
    '''Frame count of scipy.signal.stft(boundary='zeros', padded=True):
    extended length Ls + 2*(N//2); tail-padded so (L'-N) % S == 0.
    For even N with S | N this is 1 + ceil(Ls / S)  (SURVEY a1, K10).'''
    if n_samples < nperseg:
        # scipy only warns and shrinks nperseg here; the reference's fixed
        # float32 window of length N then makes _triage_segments raise.
        raise ValueError('window is longer than input signal')
    ext = n_samples + 2 * (nperseg // 2)
    nadd = (-(ext - nperseg) % stride) % nperseg
    return (ext + nadd - nperseg) // stride + 1


def stft(x, window, nperseg, stride, out_dtype='complex64', dtype=np.float64):
    '''
    scipy.signal.stft(x, window=window, nperseg=N, noverlap=N-S)[2]
        .astype(COMPLEXX).T                      (app/utils.py:117-122)
    scipy defaults: fs=1, nfft=N, detrend=False, return_onesided=True,
    boundary='zeros', padded=True, scaling='spectrum' (mode 'stft' =>
    multiply by 1/sum(window)).
    Returns complex[T, F], F = N//2+1.
    '''
    x = np.asarray(x)
    n = x.shape[-1]
    T = stft_frame_count(n, nperseg, stride)
    half = nperseg // 2
    total = nperseg + (T - 1) * stride
    ext = np.zeros(total, dtype=dtype)
    ext[half:half + n] = x.astype(dtype)
    win = np.asarray(window).astype(dtype)
    # integer framing: frame t covers ext[t*S : t*S+N]
    idx = np.arange(nperseg)[None, :] + stride * np.arange(T)[:, None]
    frames = ext[idx] * win[None, :]
    spec = np.fft.rfft(frames.astype(np.float64), n=nperseg, axis=-1)
    scale = 1.0 / float(np.asarray(window).astype(np.float64).sum())
    return (spec * scale).astype(out_dtype)


# --------------------------------------------------------------------------
# a2: iSTFT                                   (reference: app/utils.py:53-75)
# --------------------------------------------------------------------------
def istft(X, stride, window):
    '''Overlap-add inverse; float64 accumulators; uses only the frames that
    fit `range(0, len(x)-fftsize, stride)`; never undoes the 1/sum(w) scale.'''
    X = np.asarray(X)
    fftsize = (X.shape[1] - 1) * 2
    x = np.zeros(X.shape[0] * stride)
    wsum = np.zeros(X.shape[0] * stride)
    window = np.asarray(window)
    for n, i in enumerate(range(0, len(x) - fftsize, stride)):
        x[i:i + fftsize] += np.real(np.fft.irfft(X[n])) * window
        wsum[i:i + fftsize] += window ** 2.
    pos = wsum != 0
    x[pos] /= wsum[pos]
    return x


def istft_num_frames_used(T, fftsize, stride):
    '''how many leading frames app/utils.py:70 consumes'''
    return len(range(0, T * stride - fftsize, stride))


# --------------------------------------------------------------------------
# G4: random_zeropad                          (reference: app/utils.py:78-92)
# --------------------------------------------------------------------------
def random_zeropad(X, padlen, axis=-1):
    if padlen == 0:
        return X
    l = random.randint(0, padlen)
    r = padlen - l
    ndim = X.ndim
    assert -ndim <= axis < ndim
    axis %= X.ndim
    pad = [(0, 0)] * axis + [(l, r)] + [(0, 0)] * (ndim - axis - 1)
    return np.pad(X, pad, mode='constant')


# --------------------------------------------------------------------------
# a3: in-graph front-end                          (reference: main.py:233-240)
# --------------------------------------------------------------------------
def frontend(src, dtype=np.float64):
    '''src complex[B,C,T,F] -> dict(mix, src_pwr, phase, mix_pwr, mix_log)'''
    cdt = np.complex128 if dtype == np.float64 else np.complex64
    src = np.asarray(src).astype(cdt)
    mix = src.sum(axis=1)                                   # main.py:233-234
    src_pwr = np.abs(src).astype(dtype)                     # main.py:236
    phase = np.arctan2(mix.imag, mix.real).astype(dtype)    # main.py:237-238
    mix_pwr = np.abs(mix).astype(dtype)                     # main.py:239
    mix_log = np.log1p(mix_pwr).astype(dtype)               # main.py:240
    return dict(mix=mix, src_pwr=src_pwr, phase=phase,
                mix_pwr=mix_pwr, mix_log=mix_log)


# --------------------------------------------------------------------------
# a7: LSTM cell                                  (reference: app/ops.py:110-148)
# --------------------------------------------------------------------------
def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def lstm_cell(x_t, c, h, W, b):
    '''one tf.scan step.  W[D+H, 4H] rows 0..D-1 input, D.. recurrent
    (concat order app/ops.py:139); column blocks [g | i | f | o]
    (app/ops.py:143-145); g is LINEAR (no tanh).'''
    H = c.shape[-1]
    a = np.concatenate([x_t, h], axis=-1) @ W + b           # ops.py:139-142
    g = a[..., 0:H]                                         # ops.py:143
    i = sigmoid(a[..., H:2 * H])                            # ops.py:144-145
    f = sigmoid(a[..., 2 * H:3 * H])
    o = sigmoid(a[..., 3 * H:4 * H])
    c1 = i * g + f * c                                      # ops.py:146
    h1 = o * np.tanh(c1)                                    # ops.py:147
    return c1, h1


# --------------------------------------------------------------------------
# a6: scan driver                                  (reference: main.py:76-132)
# --------------------------------------------------------------------------
def lyr_lstm(x, W, b, hdim, return_cell=False):
    '''x[B,T,D] -> hid seq [B,T,H]; zero initial state (main.py:108-123),
    one cell evaluation per time step in order (tf.scan, main.py:130-131).'''
    B, T, D = x.shape
    c = np.zeros((B, hdim), dtype=x.dtype)
    h = np.zeros((B, hdim), dtype=x.dtype)
    out = np.zeros((B, T, hdim), dtype=x.dtype)
    cells = np.zeros((B, T, hdim), dtype=x.dtype)
    for t in range(T):
        c, h = lstm_cell(x[:, t], c, h, W, b)
        out[:, t] = h
        cells[:, t] = c
    if return_cell:
        return out, cells
    return out


# --------------------------------------------------------------------------
# a5: BiLSTM layer                            (reference: app/modules.py:120-137)
# --------------------------------------------------------------------------
def lyr_bilstm(x, Wf, bf, Wb, bb, hdim):
    fwd = lyr_lstm(x, Wf, bf, hdim)                         # modules.py:129-131
    bwd = lyr_lstm(x[:, ::-1], Wb, bb, hdim)                # modules.py:132-134
    # modules.py:135-137; dropout keep_prob is always 1 (main.py:243)
    return np.concatenate([fwd, bwd[:, ::-1]], axis=-1)


# --------------------------------------------------------------------------
# a4: bilstm-orig / lstm-orig encoders     (reference: app/modules.py:140-260)
# --------------------------------------------------------------------------
def lstm_bias_init(hdim, dtype=np.float64):
    '''app/modules.py:217-220'''
    b = np.zeros(4 * hdim, dtype=dtype)
    b[hdim:2 * hdim] = 1.5
    b[2 * hdim:3 * hdim] = -1.0
    b[3 * hdim:4 * hdim] = 1.0
    return b


def init_bilstm_params(rng, F, E, H=300, L=4, dtype=np.float64,
                       bidirectional=True):
    '''Initialisers of app/modules.py:212-221,248-254 (bilstm-orig) and
    :153-162,184-190 (lstm-orig).  Returns dict keyed by the reference's TF
    variable names (scopes: main.py:229, app/modules.py:208,130-133,
    app/ops.py:58-62,138).'''
    p = {}
    r = (0.75 if bidirectional else 1.15) / math.sqrt(H)
    D = F
    for l in range(L):
        dirs = ('_fwd', '_bwd') if bidirectional else ('',)
        for d in dirs:
            base = 'global/encoder/lstm%d%s/LSTM/linear/' % (l, d)
            p[base + 'W'] = rng.uniform(-r, r, size=(D + H, 4 * H)).astype(dtype)
            p[base + 'B'] = lstm_bias_init(H, dtype)
        D = 2 * H if bidirectional else H
    p['global/encoder/output/W'] = rng.uniform(
        -1.85, 1.85, size=(D, F * E)).astype(dtype)
    return p


def bilstm_encoder(x, params, H, L, E, return_all=False):
    '''BiLstmEncoder.__call__ (app/modules.py:207-260). x[B,T,F]->[B,T,F,E]'''
    B, T, F = x.shape
    x = x - x.mean(axis=(1, 2), keepdims=True)              # modules.py:209-210
    acts = []
    for l in range(L):                                      # modules.py:223-242
        x = lyr_bilstm(
            x,
            params['global/encoder/lstm%d_fwd/LSTM/linear/W' % l],
            params['global/encoder/lstm%d_fwd/LSTM/linear/B' % l],
            params['global/encoder/lstm%d_bwd/LSTM/linear/W' % l],
            params['global/encoder/lstm%d_bwd/LSTM/linear/B' % l], H)
        acts.append(x)
    y = x - x.mean(axis=(1, 2), keepdims=True)              # modules.py:244-245
    out = y @ params['global/encoder/output/W']             # modules.py:249-255
    out = out.reshape(B, -1, F, E)                          # modules.py:256-259
    if return_all:
        return out, acts, y
    return out


def lstm_encoder(x, params, H, L, E):
    '''LstmEncoder.__call__ (app/modules.py:148-196), unidirectional.'''
    B, T, F = x.shape
    x = x - x.mean(axis=(1, 2), keepdims=True)              # modules.py:150-151
    for l in range(L):                                      # modules.py:164-179
        x = lyr_lstm(
            x, params['global/encoder/lstm%d/LSTM/linear/W' % l],
            params['global/encoder/lstm%d/LSTM/linear/B' % l], H)
    y = x - x.mean(axis=(1, 2), keepdims=True)              # modules.py:181-182
    out = y @ params['global/encoder/output/W']             # modules.py:185-191
    return out.reshape(B, -1, F, E)                         # modules.py:192-195


def relu(x, alpha=0.):
    '''app/ops.py:93-107'''
    if alpha == 0.:
        return np.maximum(x, 0)
    return np.maximum(x * alpha, x)


def toy_encoder(x, params, F, E, fft_size, relu_leak):
    '''ToyEncoder (app/modules.py:104-116): two linears with bias.'''
    B = x.shape[0]
    m = x @ params['global/encoder/linear0/W'] + params['global/encoder/linear0/B']
    m = relu(m, relu_leak)
    o = m @ params['global/encoder/linear1/W'] + params['global/encoder/linear1/B']
    return o.reshape(B, -1, F, E)


# --------------------------------------------------------------------------
# a8-a10: truth estimators                 (reference: app/modules.py:382-487)
# --------------------------------------------------------------------------
def _segsum(data, idx, C):
    '''tf.unsorted_segment_sum over axis 0 of data[N, ...] with ids[N]'''
    out = np.zeros((C,) + data.shape[1:], dtype=data.dtype)
    np.add.at(out, idx, data)
    return out


def est_truth(embed, src_pwr, mix_pwr=None):
    '''AverageEstimator (app/modules.py:390-412): attr = sum / (count + 1)'''
    B, T, F, E = embed.shape
    C = src_pwr.shape[1]
    ef = embed.reshape(B, -1, E)
    idx = np.argmax(src_pwr, axis=1).reshape(B, -1)         # modules.py:396-399
    out = np.zeros((B, C, E), dtype=embed.dtype)
    for b in range(B):                                      # tf.map_fn over B
        s = _segsum(ef[b], idx[b], C)                       # modules.py:400-403
        w = _segsum(np.ones_like(ef[b]), idx[b], C)         # modules.py:404-406
        out[b] = s / (w + 1.)                               # modules.py:407
    return out


def est_truth_threshold(embed, src_pwr, mix_pwr, eps=1e-7):
    '''ThreshouldedAverageEstimator (app/modules.py:425-450): w = (5 < |mix|)'''
    B, T, F, E = embed.shape
    C = src_pwr.shape[1]
    ef = embed.reshape(B, -1, E)
    wgt = mix_pwr.reshape(B, -1, 1)                         # modules.py:431-432
    wgt = (5. < wgt).astype(embed.dtype)                    # modules.py:433-434
    idx = np.argmax(src_pwr, axis=1).reshape(B, -1)         # modules.py:435-438
    out = np.zeros((B, C, E), dtype=embed.dtype)
    for b in range(B):
        s = _segsum(ef[b] * wgt[b], idx[b], C)              # modules.py:441-443
        w = _segsum(wgt[b], idx[b], C)                      # modules.py:444-446
        out[b] = s / (w + embed.dtype.type(eps))            # modules.py:447
    return out


def est_truth_weighted(embed, src_pwr, mix_pwr, eps=1e-7):
    '''WeightedAverageEstimator (app/modules.py:462-487): w = |mix|'''
    B, T, F, E = embed.shape
    C = src_pwr.shape[1]
    ef = embed.reshape(B, -1, E)
    wgt = mix_pwr.reshape(B, -1, 1).astype(embed.dtype)     # modules.py:468-469
    idx = np.argmax(src_pwr, axis=1).reshape(B, -1)         # modules.py:470-473
    out = np.zeros((B, C, E), dtype=embed.dtype)
    for b in range(B):
        s = _segsum(ef[b] * wgt[b], idx[b], C)              # modules.py:476-478
        w = _segsum(wgt[b], idx[b], C)                      # modules.py:479-481
        out[b] = s / (w + embed.dtype.type(eps))            # modules.py:482
    return out


# --------------------------------------------------------------------------
# a11: anchor estimator      (reference: app/modules.py:490-545, ops.py:273-292)
# --------------------------------------------------------------------------
def combinations(n, k):
    '''ops.combinations index table: itertools lexicographic (ops.py:287-292)'''
    return np.asarray(list(itertools.combinations(range(n), k)), dtype=np.int64)


def softmax(x, axis=-1):
    m = x.max(axis=axis, keepdims=True)
    e = np.exp(x - m)
    return e / e.sum(axis=axis, keepdims=True)


def est_anchor(embed, anchors, C, return_all=False):
    '''AnchoredEstimator.__call__ (app/modules.py:501-545)'''
    B, T, F, E = embed.shape
    A = anchors.shape[0]
    combs = combinations(A, C)                              # modules.py:509-510
    sets = anchors[combs]                                   # [P,C,E]
    logit = np.einsum('btfe,pce->bptfc', embed, sets)       # modules.py:513-515
    assign = softmax(logit, axis=-1)                        # modules.py:516
    asets = np.einsum('bptfc,btfe->bpce', assign, embed)    # modules.py:519-521
    asets = asets / assign.sum(axis=(2, 3))[..., None]      # modules.py:522-523
    gram = asets @ np.swapaxes(asets, -1, -2)               # modules.py:527-529
    sim = gram.max(axis=(-1, -2))                           # modules.py:526,530
    choice = np.argmin(sim, axis=1)                         # modules.py:533
    attr = asets[np.arange(B), choice]                      # modules.py:534-537
    if return_all:
        return attr, dict(asets=asets, subset_choice=choice, sim=sim,
                          assign=assign)
    return attr


# --------------------------------------------------------------------------
# a12: separators                          (reference: app/modules.py:548-603)
# --------------------------------------------------------------------------
def sep_dot(mix_pwr, attr, embed_flat, act, return_masks=False):
    B, T, F = mix_pwr.shape
    C = attr.shape[1]
    logits = embed_flat @ np.swapaxes(attr, 1, 2)           # modules.py:558-560
    logits = logits.reshape(B, -1, F, C)                    # modules.py:561-565
    if act == 'softmax':
        masks = softmax(logits, axis=-1)                    # modules.py:595
    else:
        masks = sigmoid(logits)                             # modules.py:566
    sep = mix_pwr[..., None] * masks                        # modules.py:567-568
    out = np.transpose(sep, (0, 3, 1, 2))                   # modules.py:573-574
    if return_masks:
        return out, masks
    return out


def sep_softmax(mix_pwr, attr, embed_flat, **kw):
    return sep_dot(mix_pwr, attr, embed_flat, 'softmax', **kw)


def sep_sigmoid(mix_pwr, attr, embed_flat, **kw):
    return sep_dot(mix_pwr, attr, embed_flat, 'sigmoid', **kw)


# --------------------------------------------------------------------------
# a13: phase re-attach                           (reference: main.py:281-284)
# --------------------------------------------------------------------------
def reattach_phase(sep_pwr, phase):
    ph = phase[:, None]
    return (np.cos(ph) * sep_pwr) + 1j * (np.sin(ph) * sep_pwr)


# --------------------------------------------------------------------------
# a14: PIT MSE loss                              (reference: app/ops.py:374-431)
# --------------------------------------------------------------------------
def permutations(C):
    return np.asarray(list(itertools.permutations(range(C))), dtype=np.int64)


def pit_mse_loss(x, y):
    '''x (truth) , y (estimate): [B,C,T,F] complex or real.
    Returns (loss scalar, perms[P!,C], idx[B], loss_sets[B,P!])'''
    B, C = x.shape[:2]
    perms = permutations(C)                                 # ops.py:406-408
    d = x[:, :, None] - y[:, None, :]                       # ops.py:412-415
    if np.iscomplexobj(x) and np.iscomplexobj(y):
        sq = d.real ** 2 + d.imag ** 2                      # ops.py:416-418
    else:
        sq = d ** 2                                         # ops.py:420-421
    cross = sq.mean(axis=(3, 4))                            # [B,C,C]
    onehot = np.zeros((len(perms), C, C), dtype=cross.dtype)
    for p, perm in enumerate(perms):                        # ops.py:409-410
        onehot[p, np.arange(C), perm] = 1
    loss_sets = np.einsum('bij,pij->bp', cross, onehot)     # ops.py:422-423
    idx = np.argmin(loss_sets, axis=1)                      # ops.py:424
    loss = loss_sets[np.arange(B), idx].mean()              # ops.py:425-430
    return loss, perms, idx, loss_sets


def perm_gather(sep, perms, idx):
    '''main.py:293-306: out[b,c] = sep[b, perms[idx[b]][c]]'''
    B = sep.shape[0]
    return sep[np.arange(B)[:, None], perms[idx]]


# --------------------------------------------------------------------------
# a15: SNR                                       (reference: app/ops.py:191-222)
# --------------------------------------------------------------------------
def batch_snr(clear, noisy, eps=1e-7):
    noise = clear - noisy
    if np.iscomplexobj(clear) and np.iscomplexobj(noisy):
        clear = np.abs(clear)                               # ops.py:208-210
        noise = np.abs(noise)
    axes = tuple(range(1, clear.ndim))
    sp = (clear ** 2).mean(axis=axes)                       # ops.py:213-216
    npw = (noise ** 2).mean(axis=axes)
    return 4.342944819 * (np.log(sp + eps) - np.log(npw + eps))   # ops.py:221-222


# --------------------------------------------------------------------------
# Model.build() forward graph                    (reference: main.py:208-337)
# --------------------------------------------------------------------------
ESTIMATORS = {
    'truth': est_truth,
    'truth-threshold': est_truth_threshold,
    'truth-weighted': est_truth_weighted,
}


def model_forward(src, params, cfg, dtype=np.float64):
    '''
    cfg: dict(H, L, E, C, A, train_est, infer_est, separator, eps,
              encoder='bilstm-orig')
    params: encoder params + optional
            'global/train_estimator/anchors', 'global/infer_estimator/anchors'
    Returns dict of every debug_fetches-style intermediate.
    '''
    fe = frontend(src, dtype)
    H, L, E, C = cfg['H'], cfg['L'], cfg['E'], cfg['C']
    eps = cfg.get('eps', 1e-7)
    p = {k: np.asarray(v).astype(dtype) for k, v in params.items()}
    enc = cfg.get('encoder', 'bilstm-orig')
    if enc == 'bilstm-orig':
        embed = bilstm_encoder(fe['mix_log'], p, H, L, E)   # main.py:243
    elif enc == 'lstm-orig':
        embed = lstm_encoder(fe['mix_log'], p, H, L, E)
    elif enc == 'toy':                                      # app/modules.py:96-116
        F_ = fe['mix_log'].shape[-1]
        embed = toy_encoder(fe['mix_log'], p, F_, E, cfg['fft_size'], cfg.get('relu_leak', 0.))
    else:
        raise KeyError(enc)
    B, T, F, _ = embed.shape
    embed_flat = embed.reshape(B, -1, E)                    # main.py:244-246

    def run_est(name, scope):
        if name == 'anchor':
            return est_anchor(embed, p['global/%s/anchors' % scope], C)
        if name == 'truth':
            return est_truth(embed, fe['src_pwr'])
        return ESTIMATORS[name](embed, fe['src_pwr'], fe['mix_pwr'], eps)

    attrs = run_est(cfg['train_est'], 'train_estimator')    # main.py:249-254
    if cfg['infer_est'] == cfg['train_est']:                # main.py:256-267
        vattrs = attrs
    else:
        vattrs = run_est(cfg['infer_est'], 'infer_estimator')
    act = {'dot-softmax-orig': 'softmax', 'dot-sigmoid-orig': 'sigmoid'}[
        cfg['separator']]
    sep_pwr, masks = sep_dot(fe['mix_pwr'], attrs, embed_flat, act,
                             return_masks=True)             # main.py:269-272
    sep_pwr_v = sep_dot(fe['mix_pwr'], vattrs, embed_flat, act)  # main.py:274-278
    sep = reattach_phase(sep_pwr, fe['phase'])              # main.py:281-284
    srcc = np.asarray(src).astype(sep.dtype)
    loss, perms, idx, _ = pit_mse_loss(srcc, sep)           # main.py:289-290
    sep_perm = perm_gather(sep, perms, idx)                 # main.py:293-306
    snr = batch_snr(srcc, sep_perm, eps).mean()             # main.py:308-309
    vloss, _, vidx, _ = pit_mse_loss(fe['src_pwr'], sep_pwr_v)   # main.py:312-313
    sep_pwr_v_pit = perm_gather(sep_pwr_v, perms, vidx)     # main.py:314-328
    sep_v = reattach_phase(sep_pwr_v_pit, fe['phase'])      # main.py:330-332
    sep_infer = reattach_phase(sep_pwr_v, fe['phase'])      # main.py:333-335
    vsnr = batch_snr(srcc, sep_v, eps).mean()               # main.py:336-337
    return dict(fe, embed=embed, attrs=attrs, valid_attrs=vattrs, masks=masks,
                sep_pwr=sep_pwr, sep_pwr_valid=sep_pwr_v, output=sep_perm,
                sep=sep, loss=loss, perm_idx=idx, SNR=snr,
                valid_loss=vloss, valid_perm_idx=vidx, valid_SNR=vsnr,
                signals_infer=sep_infer)


# --------------------------------------------------------------------------
# a16 / f-3: TF1 Adam + value clip   (reference: main.py:354-363, ozers.py:15-18)
# --------------------------------------------------------------------------
def tf_adam_step(theta, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8, clip=100.0):
    '''tf.train.AdamOptimizer update (TF1 docs):
        lr_t = lr * sqrt(1-b2^t) / (1-b1^t);  m,v EMA;
        theta -= lr_t * m / (sqrt(v) + eps)        (eps OUTSIDE the root)
    preceded by clip_by_value(g, -clip, clip) (main.py:359-362).'''
    if clip is not None:
        g = np.clip(g, -clip, clip)
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    lr_t = lr * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    theta = theta - lr_t * m / (np.sqrt(v) + eps)
    return theta, m, v


# --------------------------------------------------------------------------
# synthetic inputs
# --------------------------------------------------------------------------
def toy_batch(rng, B, C, F, T=128):
    '''reference toy generator (app/datasets/dataset.py:56-58): real-valued
    uniform noise rand(B*C, 128, F) float32, reshaped [B,C,T,F] (main.py:417-421)'''
    sig = rng.rand(B * C, T, F).astype(np.float32)
    return sig.reshape(B, C, T, F).astype(np.complex64)


def speech_shaped_wave(rng, n_samples, smprate=8000, rms=1000.0, phase=0.0):
    '''white N(0,1) -> one-pole low-pass y[n]=0.95y[n-1]+x[n] -> slow 3 Hz
    amplitude envelope -> int16-like RMS (cfg 2/3).'''
    import scipy.signal
    x = rng.randn(n_samples).astype(np.float32)
    y = scipy.signal.lfilter([1.0], [1.0, -0.95], x)
    t = np.arange(n_samples) / float(smprate)
    y = y * (0.5 * (1.0 + np.sin(2 * np.pi * 3.0 * t + phase)))
    y = y * (rms / (np.sqrt(np.mean(y ** 2)) + 1e-12))
    return y.astype(np.float32)
