'''
GPU tests (run with -m gpu): the drop-in train loop (cli / feed) on the GPU.
Filed by component in round 5 (they used to live in test_gpu_round2/3/4.py; the helpers of each
former file keep a _r2 / _r3 / _r4 suffix).
'''


import pytest

pytestmark = pytest.mark.gpu


# ----------------------------------------------------------------------------
# from test_gpu_round4.py
# ----------------------------------------------------------------------------


import os


import time


import numpy as np


import pytest


import torch


from oracle import torch_ref as R


from test_gpu_fullsize import _setup, _synth, _cfg, relerr


GTOL_r4 = 2e-4


@pytest.fixture(autouse=True)
def _lstm_status_r4():
    # the float64 oracle's per-timestep products are tiny: on a 256-thread host torch's intra-op
    # pool makes them 8x SLOWER than 16 threads do (56 s vs 7 s for one cfg-2 step)
    import os
    n0 = torch.get_num_threads()
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    yield
    torch.set_num_threads(n0)
    from danet_amd import ops
    torch.cuda.synchronize()
    assert ops.lstm_status_ok(), 'persistent LSTM kernel reported a hand-off timeout'


def _oracle_step_r4(src, params, cfg):
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params.items()}
    r = R.model_forward(src.cpu().to(torch.complex128), tp, cfg)
    r['loss'].backward()
    return r, tp


def _train_step_vs_oracle_r4(hp, model, src, min_checked):
    from danet_amd import ops
    model.keep_grads = True                  # the optimiser leaves the bucket readable
    assert model.fuse_heads                  # the path bench.py times
    params = model.param_dict()              # BEFORE the step (Adam moves them)
    out = model.train_step(src)
    torch.cuda.synchronize()
    assert ops.lstm_status_ok()
    t0 = time.time()
    ref, tp = _oracle_step_r4(src, params, _cfg(hp))
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
    bad = {k: v for k, v in worst.items() if not v < GTOL_r4}
    print('worst gradient error: %s' % max(worst.items(), key=lambda kv: kv[1]).__repr__())
    assert not bad, bad
    assert checked >= min_checked, checked
    return out, ref


# ------------------------------------------------ data parallel, 2 ranks, the HIP path (one GPU)
_DP2_WORKER_r4 = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
import __graft_entry__ as g; g.load_package()
from danet_amd import dist, ops
from danet_amd.hparams import hparams
from danet_amd.model import Model
torch.cuda.set_device(0)                      # both ranks share the one GPU of the box
dev = torch.device('cuda', 0)
ops.prepare_streams(dev)
torch.distributed.init_process_group('gloo')  # (RCCL refuses two ranks on one device; the Model
rank, world = dist.rank(), dist.world_size()  #  code under test is backend-agnostic)
assert world == 2 and dist.is_dist()
hparams.reset()
hparams.load(%(hp)r)
hparams.digest()
model = Model('dp2', device=dev, seed=11).build()          # rank 0's parameters are broadcast
model.keep_grads = True
src = np.load(os.path.join(os.environ['DP_OUT'], 'src.npy'))
B = hparams.BATCH_SIZE
mine = torch.as_tensor(src[rank * B:(rank + 1) * B]).to(dev)
p0 = model.param_dict()
out = model.train_step(mine)
torch.cuda.synchronize()
model.check_status()
# the bucket now holds the SUM over the ranks (1/world is folded into the optimiser kernel)
np.savez(os.path.join(os.environ['DP_OUT'], 'rank%%d.npz' %% rank), loss=float(out['loss']),
         collectives=model.collectives_per_step(), status_tail=model.status_words().cpu().numpy(),
         **{'g:' + k: v for k, v in model.grad_dict().items()},
         **{'p0:' + k: v for k, v in p0.items()},
         **{'p1:' + k: v for k, v in model.param_dict().items()})
for _ in range(3):                            # a few more steps: replicas must stay identical
    model.train_step(mine)
torch.cuda.synchronize()
flat = model._flat.detach().clone()
other = [torch.empty_like(flat) for _ in range(2)]
torch.distributed.all_gather(other, flat)
assert torch.equal(other[0], other[1]), 'replicas drifted apart'
# a hand-off timeout on ONE rank (fault injection: workgroup 0 of its recurrent launches exits
# without publishing): the status word rides in the gradient all-reduce, so BOTH ranks must raise
# DanetHipError, at the admission of the SAME step
from danet_amd import _lib
model.check_status()
start = model.step_count
FAULT_AT = 2
raised_at = None
for i in range(12):
    if rank == 1 and i == FAULT_AT:
        _lib.set_option('lstm_fault_inject', 1)
        _lib.set_option('lstm_spin_limit', 2048)
    try:
        model.train_step(mine)
    except _lib.DanetHipError:
        raised_at = model.step_count - start
        break
    if rank == 1 and i == FAULT_AT:
        _lib.set_option('lstm_fault_inject', 0)
        _lib.set_option('lstm_spin_limit', 0)
open(os.path.join(os.environ['DP_OUT'], 'fault%%d.txt' %% rank), 'w').write(str(raised_at))
torch.cuda.synchronize()
torch.distributed.destroy_process_group()
'''


def test_train_loop_async_feed_equals_synchronous_loop_bit_for_bit(hp):
    '''cli.train_epoch (main.py:413-436): the one-batch-ahead pinned feed + deferred metric reads
    produce EXACTLY the epoch metrics and parameters of the reference's literal loop (blocking
    upload, float() of every metric every step) on the same batches and `random` stream'''
    import io
    import random
    from danet_amd import cli
    from danet_amd.model import Model

    def run(sync):
        hp.reset()
        hp.load(dict(BATCH_SIZE=8, MAX_N_SIGNAL=2, FFT_SIZE=256, FFT_STRIDE=64, EMBED_SIZE=20,
                     NUM_LSTM_LAYERS=2, LSTM_HDIM=300, NUM_ANCHOR=6, MAX_TRAIN_LEN=64,
                     ENCODER_TYPE='bilstm-orig', TRAIN_ESTIMATOR_METHOD='anchor',
                     INFER_ESTIMATOR_METHOD='anchor', SEPARATOR_TYPE='dot-softmax-orig'))
        hp.digest()
        model = Model('loop', device='cuda', seed=5).build()
        rng = np.random.RandomState(0)
        host = []
        for i in range(12):                      # ragged lengths: some cropped, some not
            T = [96, 64, 80, 50][i % 4]
            a = (rng.randn(16, T, hp.FEATURE_SIZE) + 1j * rng.randn(16, T, hp.FEATURE_SIZE))
            host.append(((30 * a).astype(np.complex64),))
        random.seed(9)
        out = io.StringIO()
        rep, n = cli.train_epoch(model, iter(host), out, sync_feed=sync)
        model.check_status()
        return rep, n, out.getvalue(), model._flat.detach().cpu().numpy().copy()

    rep_s, n_s, ticks_s, p_s = run(True)
    rep_a, n_a, ticks_a, p_a = run(False)
    assert n_s == n_a == 12 and ticks_s == ticks_a == ':' * 12
    assert list(rep_s) == list(rep_a) == ['loss', 'SNR', 'LR']
    for k in rep_s:
        assert rep_s[k] == rep_a[k], (k, rep_s[k], rep_a[k])       # bit for bit
    assert np.array_equal(p_s, p_a)
