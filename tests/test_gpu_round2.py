'''
GPU tests (-m gpu) added in round 2:
  * product robustness: hand-off status surfaced by Model.train_step, no per-step
    memory growth (2 000-step soak), gradient-reduction schedules respect stream order,
    fast (direct-gradient) backward == plain autograd backward, bench.py's spawn path;
  * `toy` encoder + LinearFn vs the oracle, Model.lyr_lstm and
    modules._lyr_bilstm called directly, variable-length batches padded by
    utils.random_zeropad through the full model vs the oracle;
  * BASELINE configs as written: cfg 4 at 4 x 600 / direction at model level, cfg 5 with the
    k-means estimator at T = 1251, BPTT vs the oracle at T = 512, Adam to 1e-5.
'''
import json
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import danet_oracle as O
from oracle import torch_ref as R

pytestmark = pytest.mark.gpu
TOL = 1e-4
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def relerr(a, b):
    a = np.asarray(a); b = np.asarray(b)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def cu(x, dtype=torch.float32):
    return torch.as_tensor(np.asarray(x)).to('cuda', dtype)


@pytest.fixture(autouse=True)
def _lstm_status():
    yield
    from danet_amd import ops
    torch.cuda.synchronize()
    assert ops.lstm_status_ok(), 'persistent LSTM kernel reported a hand-off timeout'


def _small_model(hp, seed=3, **kw):
    from danet_amd.model import Model
    base = dict(BATCH_SIZE=4, MAX_N_SIGNAL=2, FFT_SIZE=64, FFT_STRIDE=16, EMBED_SIZE=4,
                NUM_LSTM_LAYERS=2, LSTM_HDIM=16, NUM_ANCHOR=4, ENCODER_TYPE='bilstm-orig',
                TRAIN_ESTIMATOR_METHOD='anchor', INFER_ESTIMATOR_METHOD='anchor',
                SEPARATOR_TYPE='dot-softmax-orig')
    base.update(kw)
    hp.load(base)
    hp.digest()
    return Model('r2', device='cuda', seed=seed).build()


def _rand_src(hp, T, seed=0, scale=4.0):
    rng = np.random.RandomState(seed)
    B, C, F = hp.BATCH_SIZE, hp.MAX_N_SIGNAL, hp.FEATURE_SIZE
    return ((rng.randn(B, C, T, F) + 1j * rng.randn(B, C, T, F)) * scale).astype(np.complex64)


def _cfg(hp, **kw):
    d = dict(H=hp.LSTM_HDIM, L=hp.NUM_LSTM_LAYERS, E=hp.EMBED_SIZE, C=hp.MAX_N_SIGNAL,
             A=hp.NUM_ANCHOR, train_est=hp.TRAIN_ESTIMATOR_METHOD,
             infer_est=hp.INFER_ESTIMATOR_METHOD, separator=hp.SEPARATOR_TYPE,
             encoder=hp.ENCODER_TYPE)
    d.update(kw)
    return d


# ----------------------------------------------------------- product robustness
def test_train_soak_memory_flat(hp):
    '''2 000 train steps: torch.cuda.memory_allocated() does not grow (round 1 retained a
    view of every launch workspace: ~53 MB per step at cfg 2) and the status monitor holds
    ONE word per device'''
    from danet_amd import ops
    model = _small_model(hp)
    src = [torch.as_tensor(_rand_src(hp, 12, s)).cuda() for s in range(3)]
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
    model = _small_model(hp)
    src = torch.as_tensor(_rand_src(hp, 10)).cuda()
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


class _FakeWork(object):
    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


def test_reduction_schedules_respect_stream_order(hp, monkeypatch):
    '''A 1-rank RCCL all-reduce is an identity, so it cannot show a bucket that is reduced
    before its gradients are complete.  Stand-in collective with the stream semantics of
    ProcessGroupNCCL (its own stream, which waits for the caller's current stream; work.wait()
    joins it back) that DOUBLES its tensor: every schedule must then produce exactly 2 x the
    gradient; a piece doubled before a late contribution lands gives g1*2 + g2 instead.  A
    long burst of unrelated GEMMs keeps the side streams busy so that stream order, not luck,
    decides.'''
    from danet_amd import dist as ddist, ops
    from danet_amd.model import Model
    coll = torch.cuda.Stream()

    def fake_all_reduce(t, op=None, async_op=False):
        ev = torch.cuda.current_stream().record_event()
        coll.wait_event(ev)
        with torch.cuda.stream(coll):
            t.mul_(2.0)
            done = coll.record_event()
        w = _FakeWork(done)
        if not async_op:
            w.wait()
            return None
        return w

    monkeypatch.setattr(ddist, 'is_dist', lambda: True)
    monkeypatch.setattr(ddist, 'world_size', lambda: 1)
    monkeypatch.setattr(ddist.dist, 'all_reduce', fake_all_reduce)
    monkeypatch.setattr(ddist.dist, 'broadcast', lambda t, src=0: None)
    grads = {}
    for mode in ('0', 'tail', '1'):
        monkeypatch.setenv('DANET_OVERLAP_ALLREDUCE', mode)
        hp.reset()
        model = _small_model(hp, BATCH_SIZE=32, FFT_SIZE=256, FFT_STRIDE=64, EMBED_SIZE=20,
                             NUM_LSTM_LAYERS=3, LSTM_HDIM=300, NUM_ANCHOR=6)
        model.keep_grads = True
        model.set_learn_rate(0.0)
        src = torch.as_tensor(_rand_src(hp, 48, 1)).cuda()
        for _ in range(2):
            model.train_step(src)
        torch.cuda.synchronize()
        grads[mode] = {k: v.copy() for k, v in model.grad_dict().items()}
        if mode != '0':
            assert model._buckets.launched == (2 if mode == 'tail' else 2 * 4)
        del model
    for mode in ('tail', '1'):
        for k in grads['0']:
            assert np.array_equal(grads[mode][k], grads['0'][k]), (mode, k)
    # and '0' really is 2 x the plain gradient
    hp.reset()
    monkeypatch.setenv('DANET_OVERLAP_ALLREDUCE', '0')
    monkeypatch.setattr(ddist, 'is_dist', lambda: False)
    model = _small_model(hp, BATCH_SIZE=32, FFT_SIZE=256, FFT_STRIDE=64, EMBED_SIZE=20,
                         NUM_LSTM_LAYERS=3, LSTM_HDIM=300, NUM_ANCHOR=6)
    model.keep_grads = True
    model.set_learn_rate(0.0)
    src = torch.as_tensor(_rand_src(hp, 48, 1)).cuda()
    for _ in range(2):
        model.train_step(src)
    g1 = model.grad_dict()
    for k in g1:
        assert np.array_equal(2.0 * g1[k], grads['0'][k]), k


def test_fast_backward_equals_plain_autograd(hp):
    '''inside train_step the kernels add straight into the flat bucket; outside, autograd
    gets ordinary gradient tensors (torch.autograd.grad works) -- same numbers'''
    model = _small_model(hp, TRAIN_ESTIMATOR_METHOD='truth-weighted')
    src = torch.as_tensor(_rand_src(hp, 9, 2)).cuda()
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
            assert np.array_equal(fast[k], plain[k]), k
    # a stray backward outside train_step is cleared by the next train_step
    model.forward(src)['loss'].backward()
    model.train_step(src)
    again = model.grad_dict()
    for k in names:
        assert np.array_equal(again[k], fast[k]), k


def test_bench_spawn_path_one_rank():
    '''plain `python bench.py --gpus 1` with DANET_FORCE_DIST=1 re-executes itself under
    torch.distributed.run (the path `--gpus N` takes) and reports the RCCL group'''
    env = dict(os.environ, DANET_FORCE_DIST='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--steps', '4',
                          '--warmup', '1', '--no-cpu-baseline'],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res['n_gpus'] == 1 and res['rccl_ranks'] == 1 and res['value'] > 0
    assert res['allreduce_ms_standalone'] is not None
    assert res['config']['workload'].startswith('cfg2')
    assert res['roofline']['events_in_timed_region'] is True


# ------------------------------------------------------ f-4: toy encoder, LinearFn
def test_toy_encoder_model_vs_oracle(hp):
    '''the reference's DEFAULT encoder (default.json:33, app/modules.py:96-116): linear ->
    leaky relu -> linear through ops.lyr_linear / LinearFn; forward, loss and every
    parameter gradient vs the oracle'''
    model = _small_model(hp, ENCODER_TYPE='toy', TRAIN_ESTIMATOR_METHOD='truth-weighted',
                         SEPARATOR_TYPE='dot-sigmoid-orig', FFT_SIZE=32, FFT_STRIDE=8)
    src = _rand_src(hp, 7, 5)
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
    cfg = _cfg(hp, fft_size=hp.FFT_SIZE, relu_leak=hp.RELU_LEAKAGE, with_valid=True)
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


# ---------------------------------- a6 / a5 entry points called directly (main.py:76-132)
def test_model_lyr_lstm_entry_point(hp):
    '''Model.lyr_lstm (main.py:76-132): variable names, both time-axis conventions, zero
    initial state on every call, forward + gradients vs the oracle scan'''
    from danet_amd.model import Model
    from danet_amd import modules
    hp.load(dict(BATCH_SIZE=3))
    hp.digest()
    m = Model('lyr', device='cuda', seed=11)
    B, T, D, H = 3, 9, 10, 12
    rng = np.random.RandomState(4)
    x = rng.randn(B, T, D)
    w_init = modules._uniform_init(0.4)
    b_init = modules._const_init(O.lstm_bias_init(H))
    xt = cu(x).requires_grad_(True)
    y = m.lyr_lstm('enc/l0', xt, H, t_axis=-2, w_init=w_init, b_init=b_init)
    assert set(m.vars) == {'global/enc/l0/LSTM/linear/W', 'global/enc/l0/LSTM/linear/B'}
    W, b = m.vars['global/enc/l0/LSTM/linear/W'], m.vars['global/enc/l0/LSTM/linear/B']
    assert tuple(W.shape) == (D + H, 4 * H) and tuple(b.shape) == (4 * H,)
    Wn, bn = W.detach().cpu().double().numpy(), b.detach().cpu().double().numpy()
    want = O.lyr_lstm(x, Wn, bn, H)
    assert relerr(y.detach().cpu().numpy(), want) < TOL
    # time-major call (t_axis=0), same variables (get_variable reuses them)
    y2 = m.lyr_lstm('enc/l0', cu(x).transpose(0, 1).contiguous(), H, t_axis=0,
                    w_init=w_init, b_init=b_init)
    assert tuple(y2.shape) == (T, B, H)
    assert relerr(y2.transpose(0, 1).detach().cpu().numpy(), want) < TOL
    # second call starts from the zero state again (main.py:108-123, :538-540)
    y3 = m.lyr_lstm('enc/l0', xt, H, t_axis=-2, w_init=w_init, b_init=b_init)
    assert torch.equal(y3, y)
    dy = rng.randn(B, T, H)
    y.backward(cu(dy))
    xr = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    Wr = torch.tensor(Wn, requires_grad=True); br = torch.tensor(bn, requires_grad=True)
    (R.lstm_scan(xr, Wr, br, H) * torch.tensor(dy)).sum().backward()
    assert relerr(xt.grad.cpu().numpy(), xr.grad.numpy()) < TOL
    assert relerr(W.grad.cpu().numpy(), Wr.grad.numpy()) < TOL
    assert relerr(b.grad.cpu().numpy(), br.grad.numpy()) < TOL


def test_lyr_bilstm_entry_point(hp):
    '''modules._lyr_bilstm (app/modules.py:120-137): fwd scan || reversed scan, variable
    scopes <name>_fwd / <name>_bwd, vs the oracle'''
    from danet_amd.model import Model
    from danet_amd import modules
    hp.load(dict(BATCH_SIZE=2))
    hp.digest()
    m = Model('bi', device='cuda', seed=12)
    B, T, D, H = 2, 7, 6, 8
    rng = np.random.RandomState(9)
    x = rng.randn(B, T, D)
    y = modules._lyr_bilstm('encoder/lstm0', m, cu(x), H, -2, -1, modules._uniform_init(0.5),
                            modules._const_init(O.lstm_bias_init(H)), 1.)
    names = ['global/encoder/lstm0_%s/LSTM/linear/%s' % (d, v) for d in ('fwd', 'bwd') for v in 'WB']
    assert sorted(m.vars) == sorted(names)
    p = {k: m.vars[k].detach().cpu().double().numpy() for k in names}
    want = O.lyr_bilstm(x, p[names[0]], p[names[1]], p[names[2]], p[names[3]], H)
    assert tuple(y.shape) == (B, T, 2 * H)
    assert relerr(y.detach().cpu().numpy(), want) < TOL


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
        cfg = _cfg(hp, with_valid=True)
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
        ref = R.model_forward(src[:1].cpu().to(torch.complex128), params, _cfg(hp, with_valid=True))
    for k in ('embed', 'attrs', 'sep_pwr', 'valid_attrs', 'sep_pwr_valid'):
        assert relerr(out[k][:1].detach().cpu().numpy(), ref[k].numpy()) < TOL, k
    assert np.array_equal(out['perm_idx'][:1].cpu().numpy(), ref['perm_idx'].numpy())
    g = model.grad_dict()
    assert all(np.isfinite(v).all() for v in g.values())
    assert np.abs(g['global/encoder/lstm0_fwd/LSTM/linear/W']).max() > 0
    l0 = float(model.train_step(src)['loss'])
    l1 = float(model.train_step(src)['loss'])
    assert np.isfinite(l0) and np.isfinite(l1)


def test_cfg5_kmeans_at_full_length(hp):
    '''cfg 5 with the k-means estimator at T = 1251 (16 kHz, FFT 512/128, B = 1).  The
    estimator is an extension (README.md:216: not in the reference), so it is checked
    against the oracle's float64 restatement (oracle/torch_ref.py est_kmeans) of what
    modules.KMeansEstimator documents: start from
    the anchor estimator's attractors, KMEANS_ITERS times assign every bin to the attractor
    with the largest dot product and recompute |mix|-weighted means.  On an embedding with
    two well-separated clusters that ends at the weighted cluster means whenever both
    clusters received a start attractor; through the whole model the masks sum to one.'''
    from danet_amd.model import Model
    from danet_amd import datasets, utils, ops
    hp.load(dict(BATCH_SIZE=1, MAX_N_SIGNAL=2, FFT_SIZE=512, FFT_STRIDE=128, SMPRATE=16000,
                 EMBED_SIZE=20, NUM_LSTM_LAYERS=4, LSTM_HDIM=300, NUM_ANCHOR=6,
                 ENCODER_TYPE='bilstm-orig', TRAIN_ESTIMATOR_METHOD='truth-weighted',
                 INFER_ESTIMATOR_METHOD='kmeans', SEPARATOR_TYPE='dot-softmax-orig'))
    hp.digest()
    model = Model('c5k', device='cuda', seed=7).build()
    T, F, E = 1251, 257, 20
    rng = np.random.RandomState(3)
    centres = rng.randn(2, E) * 2.0
    assign = rng.randint(0, 2, size=(1, T, F))
    emb = (centres[assign] + 0.05 * rng.randn(1, T, F, E)).astype(np.float32)
    w = (np.abs(rng.randn(1, T, F)) + 0.1).astype(np.float32)
    got = model.valid_estimator(cu(emb), s_mix_pwr=cu(w)).cpu().numpy()[0]      # [2, E]
    anchors = model.vars['global/infer_estimator/anchors']
    a_attr, _, _ = ops.AnchorAttractorFn.apply(cu(emb), anchors.detach(), 2)
    # the oracle's float64 restatement of the extension (oracle/torch_ref.py est_kmeans)
    from oracle import torch_ref as R
    attr = R.est_kmeans(torch.tensor(emb, dtype=torch.float64),
                        anchors.detach().cpu().double(), 2, torch.tensor(w, dtype=torch.float64),
                        int(hp.KMEANS_ITERS), float(hp.EPS))[0].numpy()
    ef, wf = emb.reshape(-1, E).astype(np.float64), w.reshape(-1).astype(np.float64)
    assert relerr(got, attr) < 1e-4
    truth = np.stack([(ef[assign.reshape(-1) == c] * wf[assign.reshape(-1) == c][:, None]).sum(0)
                      / (wf[assign.reshape(-1) == c].sum() + hp.EPS) for c in range(2)])
    if len(set(np.argmax(truth @ a_attr.cpu().numpy()[0].T.astype(np.float64), axis=1))) == 2:
        order = [0, 1] if np.abs(got[0] - truth[0]).sum() < np.abs(got[0] - truth[1]).sum() else [1, 0]
        assert relerr(got, truth[order]) < 1e-3
    # whole inference chain on a real-length utterance
    w1 = datasets.speech_shaped_wave(rng, 160000, 16000, phase=0.3)
    w2 = datasets.speech_shaped_wave(rng, 160000, 16000, phase=2.1)
    X = utils.stft(torch.as_tensor((w1 + w2).astype(np.float32)).cuda())
    assert tuple(X.shape) == (T, F)
    sep = model.infer(X[None])
    assert tuple(sep.shape) == (1, 2, T, F) and bool(torch.isfinite(torch.view_as_real(sep)).all())
    assert relerr(sep.sum(1)[0].cpu().numpy(), X.cpu().numpy()) < 1e-5


def test_bptt_vs_oracle_long_sequence():
    '''T = 512: bounds what the phase bit in the LSB of the exchanged partial dh
    (csrc/lstm.hip, lstm_bwd_rs_kernel) does to gradients over long sequences -- forward
    and every gradient vs the float64 oracle, same 1e-4 bar'''
    from danet_amd import ops
    B, T, D, H = 16, 512, 24, 300
    rng = np.random.RandomState(77)
    r = 0.75 / np.sqrt(H)
    x = rng.randn(B, T, D) * 0.7
    Ws = [rng.uniform(-r, r, size=(D + H, 4 * H)) for _ in range(2)]
    bs = [O.lstm_bias_init(H) for _ in range(2)]
    dy = rng.randn(B, T, 2 * H)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    Wt = [torch.tensor(W, dtype=torch.float64, requires_grad=True) for W in Ws]
    bt = [torch.tensor(b, dtype=torch.float64, requires_grad=True) for b in bs]
    yr = torch.cat([R.lstm_scan(xt, Wt[0], bt[0], H), R.lstm_scan(xt, Wt[1], bt[1], H, reverse=True)], -1)
    (yr * torch.tensor(dy)).sum().backward()
    xc = cu(x).requires_grad_(True)
    params = []
    for W, b in zip(Ws, bs):
        params += [cu(W).requires_grad_(True), cu(b).requires_grad_(True)]
    y = ops.LstmLayerFn.apply(xc, H, *params)
    assert relerr(y.detach().cpu().numpy(), yr.detach().numpy()) < TOL
    y.backward(cu(dy))
    assert relerr(xc.grad.cpu().numpy(), xt.grad.numpy()) < TOL
    for d in range(2):
        assert relerr(params[2 * d].grad.cpu().numpy(), Wt[d].grad.numpy()) < TOL
        assert relerr(params[2 * d + 1].grad.cpu().numpy(), bt[d].grad.numpy()) < TOL


def test_adam_parameters_to_1e5(hp):
    '''three optimiser steps vs the oracle's TF1 Adam: every parameter within 1e-5 relative
    (to the parameter tensor's max) -- an epsilon inside the root or a missing bias
    correction moves the first steps by O(LR) = 3e-4 relative, 30 x the bar'''
    hp.load(dict(LR=3e-4))
    model = _small_model(hp, BATCH_SIZE=2, FFT_SIZE=16, FFT_STRIDE=4, EMBED_SIZE=3,
                         NUM_LSTM_LAYERS=1, LSTM_HDIM=4, TRAIN_ESTIMATOR_METHOD='truth-weighted',
                         SEPARATOR_TYPE='dot-sigmoid-orig')
    src = _rand_src(hp, 6, 8, scale=6.0)
    cfg = _cfg(hp)
    p0 = model.param_dict()
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in p0.items()}
    m = {k: torch.zeros_like(v) for k, v in tp.items()}
    v = {k: torch.zeros_like(v_) for k, v_ in tp.items()}
    for t in (1, 2, 3):
        model.train_step(torch.as_tensor(src).cuda())
        for k in tp:
            tp[k].grad = None
        R.model_forward(torch.tensor(src.astype(np.complex128)), tp, cfg)['loss'].backward()
        R.tf_adam_step_(tp, {k: tp[k].grad for k in tp}, m, v, t, hp.LR, clip=hp.GRAD_CLIP_THRES)
    p3 = model.param_dict()
    for k in p0:
        want = tp[k].detach().numpy()
        if tp[k].grad is None:
            assert np.array_equal(p3[k], p0[k]), k
            continue
        moved = np.abs(want - p0[k]).max()
        assert moved > 2e-4 * np.abs(p0[k]).max() or np.abs(p0[k]).max() == 0 or k.endswith('/B')
        assert np.abs(p3[k] - want).max() <= 1e-5 * max(np.abs(want).max(), 1e-3), k
        # and tight against the distance moved: an eps-placement bug is O(moved)
        assert np.abs(p3[k] - want).max() <= 2e-2 * moved + 2e-7, k


def test_early_optimizer_step_is_the_same_update(hp):
    '''clip + Adam over everything outside the bottom encoder layer is issued on the side stream
    as soon as those gradients are final (Model._grad_ready); the parameters after three steps
    are bit-identical to the single update after backward'''
    src = None
    res = []
    for early in (True, False):
        torch.manual_seed(5)
        np.random.seed(5)
        model = _small_model(hp, BATCH_SIZE=3, FFT_SIZE=16, FFT_STRIDE=4, EMBED_SIZE=4,
                             NUM_LSTM_LAYERS=2, LSTM_HDIM=8)
        if src is None:
            src = _rand_src(hp, 6, 8, scale=6.0)
            p0 = model.param_dict()
        else:
            model.load_param_dict(p0)
        model._early_adam = early
        for _ in range(3):
            model.train_step(torch.as_tensor(src).cuda())
        assert model.early_steps == (3 if early else 0)
        res.append(model.param_dict())
    for k in res[0]:
        assert np.array_equal(res[0][k], res[1][k]), k
        assert not np.array_equal(res[0][k], p0[k]) or k.endswith('/B') or np.abs(p0[k]).max() == 0, k


def test_adam_nan_gradient_propagates():
    '''tf.clip_by_value passes NaN through; a NaN gradient must poison the parameter (so the
    train loop's NaN-restore sees it), not become a +-clip update'''
    from danet_amd import ops
    n = 1030
    theta = torch.ones(n, device='cuda'); g = torch.full((n,), 0.5, device='cuda')
    g[7] = float('nan'); g[8] = 1e9; g[9] = -1e9
    m = torch.zeros(n, device='cuda'); v = torch.zeros(n, device='cuda')
    ops.adam_clip_step(theta, g, m, v, 0.1, clip=100.0, zero_grad=True)
    t = theta.cpu().numpy()
    assert np.isnan(t[7]) and np.isfinite(np.delete(t, 7)).all()
    assert abs(m[8].item() - 10.0) < 1e-5 and abs(m[9].item() + 10.0) < 1e-5     # clipped to +-100
    assert float(g.abs().nan_to_num().max()) == 0.0                                # zeroed after use


# ------------------------------- forward with the input projection fused into the scan
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
    (32, 20, 600, 300, 2),     # cfg 2 layers 1..L-1
    (32, 16, 132, 300, 2),     # layer-0-like width (3 k-groups per wave)
    (48, 7, 600, 300, 2),      # 3 clusters per direction: 228 workgroups
    (5, 9, 20, 36, 2), (16, 12, 64, 128, 1), (3, 1, 8, 8, 2), (20, 6, 640, 320, 1)])
@pytest.mark.parametrize('fused', ['1', '0'])
def test_lstm_forward_fused_input_projection(B, T, D, H, ndir, fused, monkeypatch):
    '''danet_lstm_fwd_fused (x_t*Wx computed inside the persistent scan, in the exchange
    wait) on one side, the hoisted-GEMM path on the other: both give the oracle's outputs and
    gradients; the envelope query decides which one runs'''
    from danet_amd import ops, _lib
    _lib.set_option('lstm_fwd_fused', int(fused))
    assert _lib.load().danet_lstm_fwd_fused_supported(T, B, H, ndir, D) == int(fused)
    rng = np.random.RandomState(B * 100 + T * 10 + H + D)
    r = 0.75 / np.sqrt(H)
    x = rng.randn(B, T, D) * 0.7
    Ws = [rng.uniform(-r, r, size=(D + H, 4 * H)) * 2 for _ in range(ndir)]
    bs = [O.lstm_bias_init(H) + rng.randn(4 * H) * 0.1 for _ in range(ndir)]
    dy = rng.randn(B, T, ndir * H)
    torch.set_num_threads(min(os.cpu_count() or 1, 16))
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


def test_lstm_fused_envelope_query(monkeypatch):
    from danet_amd import _lib
    L = _lib.load()
    L.danet_reset_options()
    assert L.danet_lstm_fwd_fused_supported(128, 32, 300, 2, 600) == 1      # default: B >= 24
    assert L.danet_lstm_fwd_fused_supported(1251, 1, 300, 2, 600) == 0      # B = 1: hoisted GEMM
    assert L.danet_lstm_bwd_db_supported(128, 32, 300, 2) == 1              # reduce-scatter BPTT
    assert L.danet_lstm_bwd_db_supported(128, 32, 302, 2) == 0              # H % 4 != 0: all-gather kernel
    _lib.set_option('lstm_fwd_fused', 1)
    assert L.danet_lstm_fwd_fused_supported(1251, 1, 300, 2, 600) == 1
    assert L.danet_lstm_fwd_fused_supported(128, 32, 300, 2, 129) == 1
    assert L.danet_lstm_fwd_fused_supported(128, 32, 600, 2, 1200) == 0     # H > 320
    assert L.danet_lstm_fwd_fused_supported(128, 32, 300, 2, 644) == 0      # D > 640
    assert L.danet_lstm_fwd_fused_supported(128, 64, 300, 2, 600) == 0      # 304 workgroups
    assert L.danet_lstm_fwd_fused_supported(128, 32, 302, 2, 600) == 0      # H % 4


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
