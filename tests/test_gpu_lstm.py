'''
GPU tests (run with -m gpu): the recurrent kernels and their entry points (Model.lyr_lstm, _lyr_bilstm, fused forward, BPTT, bias gradients).
Shared helpers: tests/gpu_helpers.py.
'''
import os
import time

import numpy as np
import pytest
import torch

from oracle import danet_oracle as O
from oracle import torch_ref as R
from gpu_helpers import TOL, check_lstm_status, cu, lstm_ref, relerr

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _lstm_status():
    yield
    check_lstm_status()


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
    ry, rdx, rdW, rdb = lstm_ref(x, Ws, bs, H, dy)
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
