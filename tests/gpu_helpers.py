'''
Shared helpers of the GPU tests (tests/test_gpu_*.py, run with -m gpu): error measure, device
conversion, small-model construction, the float64 torch-CPU references a whole train step or one
LSTM layer is compared with.  One copy, plain names (until round 6 every test file carried its own
_r2 / _r3 / _r4 suffixed copies).
'''
import os
import time

import numpy as np
import torch

from oracle import torch_ref as R

TOL = 1e-4           # the parity bar: error relative to the tensor's maximum
GTOL = 2e-4          # whole-step parameter gradients at full size against float64 autograd
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def relerr(a, b):
    '''max |a - b| / max |b| (float64)'''
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


def cu(x, dtype=torch.float32):
    return torch.as_tensor(np.asarray(x)).to('cuda', dtype)


def check_lstm_status():
    '''after a test: no persistent-LSTM launch reported a hand-off timeout'''
    from danet_amd import ops
    torch.cuda.synchronize()
    assert ops.lstm_status_ok(), 'persistent LSTM kernel reported a hand-off timeout'


class oracle_threads(object):
    '''the float64 oracle's per-timestep products are tiny: on a 256-thread host torch's intra-op
    pool makes them 8x SLOWER than 16 threads do (56 s vs 7 s for one cfg-2 step)'''

    def __enter__(self):
        self.n0 = torch.get_num_threads()
        torch.set_num_threads(min(16, os.cpu_count() or 1))
        return self

    def __exit__(self, *exc):
        torch.set_num_threads(self.n0)
        return False


def small_model(hp, seed=3, **kw):
    from danet_amd.model import Model
    base = dict(BATCH_SIZE=4, MAX_N_SIGNAL=2, FFT_SIZE=64, FFT_STRIDE=16, EMBED_SIZE=4,
                NUM_LSTM_LAYERS=2, LSTM_HDIM=16, NUM_ANCHOR=4, ENCODER_TYPE='bilstm-orig',
                TRAIN_ESTIMATOR_METHOD='anchor', INFER_ESTIMATOR_METHOD='anchor',
                SEPARATOR_TYPE='dot-softmax-orig')
    base.update(kw)
    hp.load(base)
    hp.digest()
    return Model('r2', device='cuda', seed=seed).build()


def rand_src(hp, T, seed=0, scale=4.0):
    rng = np.random.RandomState(seed)
    B, C, F = hp.BATCH_SIZE, hp.MAX_N_SIGNAL, hp.FEATURE_SIZE
    return ((rng.randn(B, C, T, F) + 1j * rng.randn(B, C, T, F)) * scale).astype(np.complex64)


def cfg_of(hp, **kw):
    '''oracle configuration of the model `hp` describes (incl. the encoder type)'''
    d = dict(H=hp.LSTM_HDIM, L=hp.NUM_LSTM_LAYERS, E=hp.EMBED_SIZE, C=hp.MAX_N_SIGNAL,
             A=hp.NUM_ANCHOR, train_est=hp.TRAIN_ESTIMATOR_METHOD,
             infer_est=hp.INFER_ESTIMATOR_METHOD, separator=hp.SEPARATOR_TYPE,
             encoder=hp.ENCODER_TYPE)
    d.update(kw)
    return d


class FakeWork(object):
    '''the `work` handle of a stand-in collective: wait() joins its stream back'''

    def __init__(self, ev):
        self.ev = ev

    def wait(self):
        torch.cuda.current_stream().wait_event(self.ev)


def lstm_ref(x, Ws, bs, H, dy):
    '''float64 torch-CPU (bi)LSTM layer: y, dx, dWs, dbs for the upstream gradient dy'''
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    Wt = [torch.tensor(W, dtype=torch.float64, requires_grad=True) for W in Ws]
    bt = [torch.tensor(b, dtype=torch.float64, requires_grad=True) for b in bs]
    outs = [R.lstm_scan(xt, Wt[0], bt[0], H)]
    if len(Ws) == 2:
        outs.append(R.lstm_scan(xt, Wt[1], bt[1], H, reverse=True))
    y = torch.cat(outs, dim=-1)
    (y * torch.tensor(dy)).sum().backward()
    return y.detach().numpy(), xt.grad.numpy(), [w.grad.numpy() for w in Wt], [b.grad.numpy() for b in bt]


def oracle_step(src, params, cfg):
    '''float64 torch-CPU forward + backward of the whole model at `params`'''
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params.items()}
    r = R.model_forward(src.cpu().to(torch.complex128), tp, cfg)
    r['loss'].backward()
    return r, tp


def train_step_vs_oracle(hp, model, src, min_checked):
    '''ONE Model.train_step (the kernel path bench.py times) against float64 autograd on the same
    mixtures: loss, SNR and every parameter gradient (main.py:354-358)'''
    from danet_amd import ops
    from test_gpu_fullsize import _cfg
    model.keep_grads = True                  # the optimiser leaves the bucket readable
    assert model.fuse_heads                  # the path bench.py times
    params = model.param_dict()              # BEFORE the step (Adam moves them)
    out = model.train_step(src)
    torch.cuda.synchronize()
    assert ops.lstm_status_ok()
    t0 = time.time()
    ref, tp = oracle_step(src, params, _cfg(hp))
    print('float64 oracle forward+backward: %.1f s' % (time.time() - t0))
    assert relerr(float(out['loss']), float(ref['loss'].detach())) < 1e-4
    assert relerr(float(out['SNR']), float(ref['SNR'].detach())) < 1e-4
    g = model.grad_dict()
    worst, checked = {}, 0
    for k in tp:
        if tp[k].grad is None:               # e.g. the inference estimator's anchors (main.py:362)
            assert not np.any(g[k]), k
            continue
        worst[k] = relerr(g[k], tp[k].grad.numpy())
        checked += 1
    bad = {k: v for k, v in worst.items() if not v < GTOL}
    print('worst gradient error: %s' % max(worst.items(), key=lambda kv: kv[1]).__repr__())
    assert not bad, bad
    assert checked >= min_checked, checked
    return out, ref
