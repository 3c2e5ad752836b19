'''
GPU tests (run with -m gpu): value clip + TF1-Adam kernel.
Shared helpers: tests/gpu_helpers.py.
'''
import numpy as np
import pytest
import torch

from oracle import torch_ref as R
from gpu_helpers import cfg_of, check_lstm_status, rand_src, small_model

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _lstm_status():
    yield
    check_lstm_status()


def test_adam_parameters_to_1e5(hp):
    '''three optimiser steps vs the oracle's TF1 Adam: every parameter within 1e-5 relative
    (to the parameter tensor's max) -- an epsilon inside the root or a missing bias
    correction moves the first steps by O(LR) = 3e-4 relative, 30 x the bar'''
    hp.load(dict(LR=3e-4))
    model = small_model(hp, BATCH_SIZE=2, FFT_SIZE=16, FFT_STRIDE=4, EMBED_SIZE=3,
                         NUM_LSTM_LAYERS=1, LSTM_HDIM=4, TRAIN_ESTIMATOR_METHOD='truth-weighted',
                         SEPARATOR_TYPE='dot-sigmoid-orig')
    src = rand_src(hp, 6, 8, scale=6.0)
    cfg = cfg_of(hp)
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
        model = small_model(hp, BATCH_SIZE=3, FFT_SIZE=16, FFT_STRIDE=4, EMBED_SIZE=4,
                             NUM_LSTM_LAYERS=2, LSTM_HDIM=8)
        if src is None:
            src = rand_src(hp, 6, 8, scale=6.0)
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
