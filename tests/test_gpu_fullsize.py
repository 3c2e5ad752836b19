'''
GPU parity at BASELINE.json's FULL sizes (run with -m gpu).

The float64 oracle cannot run a whole cfg-2/4 batch in seconds, so full-size runs
are checked through (a) the oracle on a sub-batch -- mixtures are independent
through the whole forward, and the gradient of a B=2 model is compared in full --
and (b) size-independent properties of the domain: softmax masks sum to one
(sum_c separated = mixture), batch-permutation equivariance (exercises the
batch-cluster decomposition of the persistent LSTM kernels), PIT invariance under
source swaps, BiLSTM time-reversal symmetry, linearity of the iSTFT.
'''
import numpy as np
import pytest
import torch

from oracle import danet_oracle as O
from oracle import torch_ref as R

pytestmark = pytest.mark.gpu
TOL = 1e-4


def relerr(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.fixture(autouse=True)
def _lstm_status():
    yield
    from danet_amd import ops
    torch.cuda.synchronize()
    assert ops.lstm_status_ok(), 'persistent LSTM kernel reported a hand-off timeout'


def _setup(hp, **kw):
    from danet_amd.model import Model
    base = dict(MAX_N_SIGNAL=2, FFT_SIZE=256, FFT_STRIDE=64, SMPRATE=8000, EMBED_SIZE=20,
                NUM_LSTM_LAYERS=3, LSTM_HDIM=300, NUM_ANCHOR=6, ENCODER_TYPE='bilstm-orig',
                TRAIN_ESTIMATOR_METHOD='anchor', INFER_ESTIMATOR_METHOD='anchor',
                SEPARATOR_TYPE='dot-softmax-orig')
    base.update(kw)
    hp.load(base)
    hp.digest()
    return Model('full', device='cuda', seed=7).build()


def _synth(hp, B, T, seed):
    from danet_amd import datasets, utils
    C = hp.MAX_N_SIGNAL
    waves = datasets.synth_waves(seed, B * C, T, hp.SMPRATE)
    spec = utils.stft(torch.as_tensor(waves).cuda())
    assert spec.shape[1] == T                                   # T = 1 + ceil(Ls/S), exact
    return spec.reshape(B, C, T, hp.FEATURE_SIZE).contiguous()


def _cfg(hp, **kw):
    d = dict(H=hp.LSTM_HDIM, L=hp.NUM_LSTM_LAYERS, E=hp.EMBED_SIZE, C=hp.MAX_N_SIGNAL,
             A=hp.NUM_ANCHOR, train_est=hp.TRAIN_ESTIMATOR_METHOD,
             infer_est=hp.INFER_ESTIMATOR_METHOD, separator=hp.SEPARATOR_TYPE)
    d.update(kw)
    return d


# ---------------------------------------------------------------- cfg 2 (B=32)
def test_cfg2_full_batch_properties_and_oracle_subbatch(hp):
    model = _setup(hp, BATCH_SIZE=32)
    src = _synth(hp, 32, 128, 1337)
    with torch.no_grad():
        out = model.forward(src, with_valid=True)
    # softmax masks: sum_c separated magnitude == mixture magnitude
    assert relerr(out['sep_pwr'].sum(1).cpu().numpy(), out['mix_pwr'].cpu().numpy()) < 1e-5
    assert bool(torch.isfinite(out['embed']).all())
    # oracle on the first two mixtures (independent of the other 30)
    params = {k: torch.tensor(v, dtype=torch.float64) for k, v in model.param_dict().items()}
    with torch.no_grad():
        ref = R.model_forward(src[:2].cpu().to(torch.complex128), params, _cfg(hp, with_valid=True))
    for k in ('embed', 'attrs', 'sep_pwr', 'sep_pwr_valid'):
        assert relerr(out[k][:2].cpu().numpy(), ref[k].numpy()) < TOL, k
    assert np.array_equal(out['perm_idx'][:2].cpu().numpy(), ref['perm_idx'].numpy())
    # batch-permutation equivariance at full size (rows move between LSTM clusters)
    perm = torch.randperm(32, generator=torch.Generator().manual_seed(0)).cuda()
    with torch.no_grad():
        out_p = model.forward(src[perm].contiguous())
    assert relerr(out_p['sep_pwr'].cpu().numpy(), out['sep_pwr'][perm].cpu().numpy()) < 2e-5
    assert abs(float(out_p['loss']) - float(out['loss'])) < 1e-5 * abs(float(out['loss']))
    # PIT: swapping the two sources leaves the loss unchanged and flips the index
    with torch.no_grad():
        out_s = model.forward(src.flip(1).contiguous())
    assert abs(float(out_s['loss']) - float(out['loss'])) < 1e-5 * abs(float(out['loss']))
    assert bool(((out_s['perm_idx'] + out['perm_idx']) == 1).all())


def test_cfg2_shape_full_gradient_vs_oracle_b2(hp):
    '''every parameter gradient of the cfg-2 model (T=128, 3x300, anchor) at B=2
    against float64 torch-CPU autograd'''
    model = _setup(hp, BATCH_SIZE=2)
    src = _synth(hp, 2, 128, 99)
    model._flat_grad.zero_()
    out = model.forward(src)
    out['loss'].backward()
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True)
          for k, v in model.param_dict().items()}
    r = R.model_forward(src.cpu().to(torch.complex128), tp, _cfg(hp))
    r['loss'].backward()
    assert relerr(float(out['loss'].detach()), float(r['loss'])) < TOL
    g = model.grad_dict()
    for k in tp:
        assert relerr(g[k], tp[k].grad.numpy()) < 2 * TOL, k


def test_bilstm_time_reversal_symmetry_full_size():
    '''K3 at H=300, B=32: swapping the direction weights and reversing time
    mirrors the output'''
    from danet_amd import ops
    rng = np.random.RandomState(0)
    B, T, D, H = 32, 96, 64, 300
    x = torch.as_tensor(rng.randn(B, T, D).astype(np.float32)).cuda()
    r = 0.75 / np.sqrt(H)
    Wf, Wb = [torch.as_tensor(rng.uniform(-r, r, (D + H, 4 * H)).astype(np.float32)).cuda() for _ in range(2)]
    bf, bb = [torch.as_tensor((O.lstm_bias_init(H) + 0.1 * rng.randn(4 * H)).astype(np.float32)).cuda() for _ in range(2)]
    y = ops.LstmLayerFn.apply(x, H, Wf, bf, Wb, bb)
    y2 = ops.LstmLayerFn.apply(x.flip(1).contiguous(), H, Wb, bb, Wf, bf)
    assert relerr(y2[..., H:].flip(1).cpu().numpy(), y[..., :H].cpu().numpy()) < 1e-5
    assert relerr(y2[..., :H].flip(1).cpu().numpy(), y[..., H:].cpu().numpy()) < 1e-5


# ------------------------------------------- cfg 4 (3 speakers, E=40, 4 layers)
def test_cfg4_three_speakers_truth_weighted(hp):
    model = _setup(hp, BATCH_SIZE=32, MAX_N_SIGNAL=3, EMBED_SIZE=40, NUM_LSTM_LAYERS=4,
                   TRAIN_ESTIMATOR_METHOD='truth-weighted')
    src = _synth(hp, 32, 128, 4)
    model._flat_grad.zero_()
    out = model.forward(src, with_valid=True)
    out['loss'].backward()
    assert relerr(out['sep_pwr'].detach().sum(1).cpu().numpy(), out['mix_pwr'].cpu().numpy()) < 1e-5
    params = {k: torch.tensor(v, dtype=torch.float64) for k, v in model.param_dict().items()}
    with torch.no_grad():
        ref = R.model_forward(src[:2].cpu().to(torch.complex128), params, _cfg(hp, with_valid=True))
    for k in ('embed', 'attrs', 'sep_pwr', 'valid_attrs', 'sep_pwr_valid'):
        assert relerr(out[k][:2].detach().cpu().numpy(), ref[k].numpy()) < TOL, k
    assert np.array_equal(out['perm_idx'][:2].cpu().numpy(), ref['perm_idx'].numpy())
    assert int(out['perm_idx'].max()) <= 5 and int(out['perm_idx'].min()) >= 0      # 3! perms
    g = model.grad_dict()
    assert all(np.isfinite(v).all() for v in g.values())
    assert np.all(g['global/infer_estimator/anchors'] == 0)     # never trained (main.py:362)
    # a few optimiser steps reduce the training loss on a fixed batch
    l0 = float(model.train_step(src)['loss'])
    for _ in range(4):
        l1 = float(model.train_step(src)['loss'])
    assert np.isfinite(l1) and l1 < l0


# ------------------------- cfg 5 (16 kHz, FFT 512/128, 10 s utterance, inference)
def test_cfg5_long_utterance_inference_chain(hp):
    from danet_amd import ops, utils
    model = _setup(hp, BATCH_SIZE=1, FFT_SIZE=512, FFT_STRIDE=128, SMPRATE=16000,
                   NUM_LSTM_LAYERS=4, TRAIN_ESTIMATOR_METHOD='truth-weighted')
    rng = np.random.RandomState(5)
    from danet_amd import datasets
    w1 = datasets.speech_shaped_wave(rng, 160000, 16000, phase=0.3)
    w2 = datasets.speech_shaped_wave(rng, 160000, 16000, phase=2.1)
    mix_wave = (w1 + w2).astype(np.float32)
    X = utils.stft(torch.as_tensor(mix_wave).cuda())            # [1251, 257]
    assert tuple(X.shape) == (1251, 257)
    sep = model.infer(X[None])                                  # [1, 2, 1251, 257]
    assert tuple(sep.shape) == (1, 2, 1251, 257) and bool(torch.isfinite(torch.view_as_real(sep)).all())
    # softmax masks: the separated spectra add up to the mixture spectrum
    assert relerr(sep.sum(1)[0].cpu().numpy(), X.cpu().numpy()) < 1e-5
    # ... and, by linearity of the iSTFT, so do the waveforms
    wnd = torch.as_tensor(np.asarray(hp.FFT_WND)).cuda()
    ys = ops.istft(sep[0].contiguous(), hp.FFT_STRIDE, wnd)
    ym = ops.istft(X, hp.FFT_STRIDE, wnd)
    assert tuple(ys.shape) == (2, 1251 * 128)
    assert relerr(ys.sum(0).cpu().numpy(), ym.cpu().numpy()) < 1e-5
    # oracle on the whole utterance (B=1 inference is cheap enough on the CPU)
    params = {k: torch.tensor(v, dtype=torch.float64) for k, v in model.param_dict().items()}
    with torch.no_grad():
        fe = R.frontend(X[None, None].cpu().to(torch.complex128))
        emb = R.bilstm_encoder(fe['mix_log'], params, 300, 4, 20)
        attr = R.est_anchor(emb, params['global/infer_estimator/anchors'], 2)
        sp, _ = R.sep_dot(fe['mix_pwr'], attr, emb.reshape(1, -1, 20), 'softmax')
    got = sep.abs()[0].cpu().numpy()
    assert relerr(got, sp[0].numpy()) < TOL
