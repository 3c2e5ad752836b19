'''
GPU tests (run with -m gpu): whole-model tests on the HIP path: forward / gradients / train steps against the oracle, soak, fault injection.
Shared helpers: tests/gpu_helpers.py.
'''
import os
import random
import time

import numpy as np
import pytest
import torch

from oracle import danet_oracle as O
from oracle import torch_ref as R
from gpu_helpers import (TOL, cfg_of, check_lstm_status, cu, oracle_threads, rand_src, relerr, small_model,
                         train_step_vs_oracle)
from test_gpu_fullsize import _setup, _synth, _cfg

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _lstm_status():
    with oracle_threads():
        yield
    check_lstm_status()


# ----------------------------------------------------------- product robustness
def test_train_soak_memory_flat(hp):
    '''2 000 train steps: torch.cuda.memory_allocated() does not grow (round 1 retained a
    view of every launch workspace: ~53 MB per step at cfg 2) and the status monitor holds
    ONE word per device'''
    from danet_amd import ops
    model = small_model(hp)
    src = [torch.as_tensor(rand_src(hp, 12, s)).cuda() for s in range(3)]
    for i in range(50):
        model.train_step(src[i % 3])
    torch.cuda.synchronize()
    m0 = torch.cuda.memory_allocated()
    for i in range(2000):
        out = model.train_step(src[i % 3])
    torch.cuda.synchronize()
    m1 = torch.cuda.memory_allocated()
    assert m1 <= m0 + (1 << 20), (m0, m1)
    assert np.isfinite(float(out['loss']))
    assert len(ops._status) == 1 and ops.status_word(model.device).numel() == 4
    # the host never runs more than MAX_STEPS_IN_FLIGHT steps ahead of the GPU
    assert len(ops._dev_status(model.device).queue) <= ops.MAX_STEPS_IN_FLIGHT
    assert ops._dev_status(model.device).retired >= 2000 - ops.MAX_STEPS_IN_FLIGHT
    model.check_status()


def test_injected_handoff_timeout_raises_from_train_step(hp, monkeypatch):
    '''a hand-off timeout inside a persistent LSTM launch (forced: workgroup 0 of every
    launch exits without publishing) surfaces as DanetHipError from Model.train_step
    within ops.MAX_STEPS_IN_FLIGHT steps, and the model is usable again afterwards'''
    from danet_amd import ops, _lib
    model = small_model(hp)
    src = torch.as_tensor(rand_src(hp, 10)).cuda()
    model.train_step(src)
    model.check_status()
    _lib.set_option('lstm_fault_inject', 1)
    _lib.set_option('lstm_spin_limit', 2048)
    with pytest.raises(_lib.DanetHipError, match='hand-off timed out'):
        for _ in range(ops.MAX_STEPS_IN_FLIGHT + 2):      # no synchronisation: the fence finds it
            model.train_step(src)
    _lib.set_option('lstm_fault_inject', 0)
    _lib.set_option('lstm_spin_limit', 0)
    torch.cuda.synchronize()
    ops.lstm_status_ok()                        # clear what the faulty launches left behind
    # blocking form
    _lib.set_option('lstm_fault_inject', 1)
    _lib.set_option('lstm_spin_limit', 2048)
    with torch.no_grad():
        model.forward(src)
    _lib.set_option('lstm_fault_inject', 0)
    with pytest.raises(_lib.DanetHipError):
        model.check_status()
    model.load_param_dict({k: np.where(np.isfinite(v), v, 0.0) for k, v in model.param_dict().items()})
    out = model.valid_step(src)
    torch.cuda.synchronize()
    assert np.isfinite(float(out['loss']))


def test_fast_backward_equals_plain_autograd(hp):
    '''inside train_step the kernels add straight into the flat bucket; outside, autograd
    gets ordinary gradient tensors (torch.autograd.grad works) -- same numbers'''
    model = small_model(hp, TRAIN_ESTIMATOR_METHOD='truth-weighted')
    src = torch.as_tensor(rand_src(hp, 9, 2)).cuda()
    out = model.forward(src, fuse_heads=model.fuse_heads)     # the kernels train_step runs
    names = [k for k in model._order]
    plist = [model.vars[k] for k in names]
    gs = torch.autograd.grad(out['loss'], plist, allow_unused=True)
    plain = {k: (g.cpu().numpy() if g is not None else None) for k, g in zip(names, gs)}
    assert plain['global/encoder/output/W'] is not None
    assert all(float(v.grad.abs().max()) == 0 for v in plist)       # .grad untouched
    model.keep_grads = True
    model.set_learn_rate(0.0)
    model.train_step(src)
    fast = model.grad_dict()
    for k in names:
        if plain[k] is None:
            assert np.all(fast[k] == 0), k
        else:
            # (bit-equal until round 6: inside train_step the attractor gradient now comes from the fused
            # forward's partials -- the same products and sums, contracted differently by the compiler: an ulp)
            assert np.abs(fast[k] - plain[k]).max() <= 1e-6 * np.abs(plain[k]).max(), k
    # a stray backward outside train_step is cleared by the next train_step
    model.forward(src)['loss'].backward()
    model.train_step(src)
    again = model.grad_dict()
    for k in names:
        assert np.array_equal(again[k], fast[k]), k


# ------------------------------------------------------ f-4: toy encoder, LinearFn
def test_toy_encoder_model_vs_oracle(hp):
    '''the reference's DEFAULT encoder (default.json:33, app/modules.py:96-116): linear ->
    leaky relu -> linear through ops.lyr_linear / LinearFn; forward, loss and every
    parameter gradient vs the oracle'''
    model = small_model(hp, ENCODER_TYPE='toy', TRAIN_ESTIMATOR_METHOD='truth-weighted',
                         SEPARATOR_TYPE='dot-sigmoid-orig', FFT_SIZE=32, FFT_STRIDE=8)
    src = rand_src(hp, 7, 5)
    params = model.param_dict()
    assert params['global/encoder/linear0/W'].shape == (hp.FEATURE_SIZE, 2 * hp.FFT_SIZE)
    assert params['global/encoder/linear1/B'].shape == (hp.FEATURE_SIZE * hp.EMBED_SIZE,)
    # biases start at zero: perturb so their gradient path is exercised at a generic point
    rng = np.random.RandomState(1)
    params = {k: (v + 0.1 * rng.randn(*v.shape).astype(np.float32) if k.endswith('/B') else v)
              for k, v in params.items()}
    model.load_param_dict(params)
    out = model.forward(torch.as_tensor(src).cuda(), with_valid=True)
    out['loss'].backward()
    cfg = cfg_of(hp, fft_size=hp.FFT_SIZE, relu_leak=hp.RELU_LEAKAGE, with_valid=True)
    ref = O.model_forward(src.astype(np.complex128), params, cfg)
    for k in ('embed', 'attrs', 'sep_pwr', 'sep_pwr_valid'):
        assert relerr(out[k].detach().cpu().numpy(), ref[k]) < TOL, k
    assert relerr(float(out['loss']), ref['loss']) < TOL
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params.items()}
    R.model_forward(torch.tensor(src.astype(np.complex128)), tp, cfg)['loss'].backward()
    g = model.grad_dict()
    for k in params:
        if tp[k].grad is None:
            assert np.all(g[k] == 0), k
        else:
            assert relerr(g[k], tp[k].grad.numpy()) < 2 * TOL, k


# -------------------------- f-2: variable-length batches (random_zeropad) through the model
def test_varlen_zero_padded_batch_through_full_model(hp):
    '''`synth-varlen` batches utterances of different lengths like app/datasets/wsj0.py:51-55
    (utils.random_zeropad on the time axis).  Zero-padded frames (|mix| = 0, all |src| = 0:
    the argmax / weight tie cases, K9) flow through every estimator of the full model and
    match the oracle; training on such batches runs through cli.train.'''
    from danet_amd import cli
    hp.load(dict(DATASET_TYPE='synth-varlen', BATCH_SIZE=4, MAX_N_SIGNAL=2, FFT_SIZE=64,
                 FFT_STRIDE=16, EMBED_SIZE=4, NUM_LSTM_LAYERS=1, LSTM_HDIM=8, NUM_ANCHOR=4,
                 MAX_TRAIN_LEN=None, ENCODER_TYPE='bilstm-orig', INFER_ESTIMATOR_METHOD='anchor'))
    hp.digest()
    ds = hp.get_dataset()()
    ds.N_FRAMES, ds.MIN_FRAMES = 24, 10
    ds.install_and_load()
    random.seed(5)
    batch = next(iter(ds.epoch('train', hp.BATCH_SIZE * hp.MAX_N_SIGNAL)))[0]
    assert batch.shape[0] == 8 and batch.shape[2] == hp.FEATURE_SIZE and np.iscomplexobj(batch)
    energy = np.abs(batch).sum(-1)                                   # [8, T]
    padded = (energy == 0)
    assert padded.any() and not padded.all(axis=1).any()
    for row in padded:                                              # padding only at the two ends
        nz = np.flatnonzero(~row)
        assert not row[nz[0]:nz[-1] + 1].any()
    src = batch.reshape(hp.BATCH_SIZE, hp.MAX_N_SIGNAL, -1, hp.FEATURE_SIZE).astype(np.complex64)
    # make sure at least one (mixture, frame) is padded in EVERY source: |mix| == 0 there
    src[1, :, :3] = 0
    assert (np.abs(src).sum((1, 3)) == 0).any()
    for est, sep in (('truth', 'dot-softmax-orig'), ('truth-threshold', 'dot-sigmoid-orig'),
                     ('truth-weighted', 'dot-softmax-orig'), ('anchor', 'dot-softmax-orig')):
        from danet_amd.model import Model
        hp.load(dict(TRAIN_ESTIMATOR_METHOD=est, SEPARATOR_TYPE=sep))
        model = Model('vl', device='cuda', seed=2).build()
        params = model.param_dict()
        out = model.forward(torch.as_tensor(src).cuda(), with_valid=True)
        out['loss'].backward()
        cfg = cfg_of(hp, with_valid=True)
        ref = O.model_forward(src.astype(np.complex128), params, cfg)
        for k in ('embed', 'attrs', 'sep_pwr', 'valid_attrs', 'sep_pwr_valid'):
            assert relerr(out[k].detach().cpu().numpy(), ref[k]) < TOL, (est, k)
        for k in ('loss', 'SNR', 'valid_loss'):
            assert relerr(float(out[k]), ref[k]) < TOL, (est, k)
        assert np.array_equal(out['perm_idx'].cpu().numpy(), ref['perm_idx'])
        tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params.items()}
        R.model_forward(torch.tensor(src.astype(np.complex128)), tp, cfg)['loss'].backward()
        g = model.grad_dict()
        for k in params:
            if tp[k].grad is not None:
                assert relerr(g[k], tp[k].grad.numpy()) < 2 * TOL, (est, k)
    # the train loop consumes such batches (crop disabled: MAX_TRAIN_LEN None)
    import io
    import types
    args = types.SimpleNamespace(no_save_on_epoch=True, no_valid_on_epoch=False)
    buf = io.StringIO()
    ds.N_BATCH = {'train': 3, 'valid': 1, 'test': 1}
    cli.train(model, 1, ds, args, buf)
    assert 'Epoch 1/1' in buf.getvalue() and 'nan' not in buf.getvalue().lower()


# ------------------------------------------------------- configs as BASELINE writes them
def test_cfg4_as_written_4x600_model_level(hp):
    '''cfg 4 literally: 3 speakers, E = 40, 4 x 600 units per direction, truth-weighted
    training path, B = 32: full model forward + backward on the GPU, oracle on mixture 0
    (mixtures are independent), softmax masks sum to one over the whole batch, one
    optimiser step lowers nothing to NaN'''
    from danet_amd.model import Model
    from danet_amd import datasets, utils
    hp.load(dict(BATCH_SIZE=32, MAX_N_SIGNAL=3, FFT_SIZE=256, FFT_STRIDE=64, SMPRATE=8000,
                 EMBED_SIZE=40, NUM_LSTM_LAYERS=4, LSTM_HDIM=600, NUM_ANCHOR=6,
                 ENCODER_TYPE='bilstm-orig', TRAIN_ESTIMATOR_METHOD='truth-weighted',
                 INFER_ESTIMATOR_METHOD='anchor', SEPARATOR_TYPE='dot-softmax-orig'))
    hp.digest()
    model = Model('c4', device='cuda', seed=7).build()
    T = 128
    waves = datasets.synth_waves(41, 32 * 3, T, hp.SMPRATE)
    src = utils.stft(torch.as_tensor(waves).cuda()).reshape(32, 3, T, hp.FEATURE_SIZE).contiguous()
    out = model.forward(src, with_valid=True)
    out['loss'].backward()
    assert relerr(out['sep_pwr'].detach().sum(1).cpu().numpy(), out['mix_pwr'].cpu().numpy()) < 1e-5
    params = {k: torch.tensor(v, dtype=torch.float64) for k, v in model.param_dict().items()}
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    with torch.no_grad():
        ref = R.model_forward(src[:1].cpu().to(torch.complex128), params, cfg_of(hp, with_valid=True))
    for k in ('embed', 'attrs', 'sep_pwr', 'valid_attrs', 'sep_pwr_valid'):
        assert relerr(out[k][:1].detach().cpu().numpy(), ref[k].numpy()) < TOL, k
    assert np.array_equal(out['perm_idx'][:1].cpu().numpy(), ref['perm_idx'].numpy())
    g = model.grad_dict()
    assert all(np.isfinite(v).all() for v in g.values())
    assert np.abs(g['global/encoder/lstm0_fwd/LSTM/linear/W']).max() > 0
    l0 = float(model.train_step(src)['loss'])
    l1 = float(model.train_step(src)['loss'])
    assert np.isfinite(l0) and np.isfinite(l1)


@pytest.mark.parametrize('alpha', [0.0, 0.3])
def test_leaky_relu_kernel(alpha):
    '''ops.relu (app/ops.py:93-107): values and gradient incl. exact zeros'''
    from danet_amd import ops
    rng = np.random.RandomState(2)
    x = rng.randn(3, 37, 11).astype(np.float32)
    x[0, 0, :4] = 0.0
    dy = rng.randn(3, 37, 11).astype(np.float32)
    xt = cu(x).requires_grad_(True)
    y = ops.relu(xt, alpha)
    assert np.array_equal(y.detach().cpu().numpy(), O.relu(x, alpha).astype(np.float32))
    y.backward(cu(dy))
    assert np.array_equal(xt.grad.cpu().numpy(), dy * np.where(x > 0, 1.0, alpha).astype(np.float32))


# ------------------------------------------------ data parallel, 2 ranks, the HIP path (one GPU)


def test_cfg2_b32_train_step_gradients_vs_oracle(hp):
    '''BASELINE configs[1] exactly as bench.py runs it: B = 32, T = 128, 3 x 300, anchor
    estimator, dot-softmax.  13 BiLSTM tensors + W_out + anchors.'''
    from danet_amd import _lib
    model = _setup(hp, BATCH_SIZE=32)
    L = _lib.load()
    # the kernels the timed step takes at this shape
    assert L.danet_lstm_fwd_fused_supported(128, 32, 300, 2, 132) == 1
    assert L.danet_lstm_fwd_fused_supported(128, 32, 300, 2, 600) == 1
    assert L.danet_lstm_bwd_db_supported(128, 32, 300, 2) == 1
    src = _synth(hp, 32, 128, 1337)
    out, ref = train_step_vs_oracle(hp, model, src, min_checked=14)
    # (the fused path returns the permutation through the side-stream finalizer)
    with torch.no_grad():
        o2 = model.forward(src, fuse_heads=True)        # parameters have moved: only a smoke check
    assert int(o2['perm_idx'].min()) >= 0 and int(o2['perm_idx'].max()) <= 1


def test_cfg2_b32_perm_idx_and_trajectory_vs_oracle(hp):
    '''permutation indices of the fused path at B = 32 (read before the optimiser moves anything:
    forward only), then THREE train steps against three float64 TF1-Adam steps of the oracle
    (main.py:359-363): the loss trajectory must agree'''
    model = _setup(hp, BATCH_SIZE=32)
    src = _synth(hp, 32, 128, 2024)
    params = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True)
              for k, v in model.param_dict().items()}
    with torch.no_grad():
        o = model.forward(src, fuse_heads=True)
        r0 = R.model_forward(src.cpu().to(torch.complex128), params, _cfg(hp))
    assert np.array_equal(o['perm_idx'].cpu().numpy(), r0['perm_idx'].numpy())
    m = {k: torch.zeros_like(v) for k, v in params.items()}
    v = {k: torch.zeros_like(p) for k, p in params.items()}
    for t in range(1, 4):
        got = float(model.train_step(src)['loss'])
        for p in params.values():
            p.grad = None
        r = R.model_forward(src.cpu().to(torch.complex128), params, _cfg(hp))
        r['loss'].backward()
        assert relerr(got, float(r['loss'])) < 2e-4, (t, got, float(r['loss']))
        R.tf_adam_step_(params, {k: p.grad for k, p in params.items()}, m, v, t, float(hp.LR),
                        clip=float(hp.GRAD_CLIP_THRES))


def test_cfg4_b32_train_step_gradients_vs_oracle(hp):
    '''BASELINE configs[3] at H = 300: C = 3, E = 40, L = 4, truth-weighted training estimator
    (6 permutations; the separator-term recompute runs inside danet_attractor_truth_bwd_sep)'''
    model = _setup(hp, BATCH_SIZE=32, MAX_N_SIGNAL=3, EMBED_SIZE=40, NUM_LSTM_LAYERS=4,
                   TRAIN_ESTIMATOR_METHOD='truth-weighted')
    src = _synth(hp, 32, 128, 4)
    out, ref = train_step_vs_oracle(hp, model, src, min_checked=17)


def test_cfg4_h600_b32_train_step_gradients_vs_oracle(hp):
    '''BASELINE configs[3] as written (4 x 600): hoisted input GEMM + persistent forward, BPTT at
    H = 600, weight-gradient groups serial on the main stream.  Every layer's gradients (bottom
    and top included) against the oracle.'''
    model = _setup(hp, BATCH_SIZE=32, MAX_N_SIGNAL=3, EMBED_SIZE=40, NUM_LSTM_LAYERS=4,
                   LSTM_HDIM=600, TRAIN_ESTIMATOR_METHOD='truth-weighted')
    src = _synth(hp, 32, 128, 6)
    out, ref = train_step_vs_oracle(hp, model, src, min_checked=17)
