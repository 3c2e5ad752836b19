'''
Round 4 (run with -m gpu): the oracle next to the EXACT kernel path bench.py times, at full size.

`Model.train_step` at BASELINE cfg 2 / cfg 4 with B = 32, T = 128 takes: the fused-input
persistent forward (B >= 24, H <= 384) or the hoisted GEMM + persistent forward (H = 600), the
reduce-scatter BPTT with twins and in-kernel bias sums, the stream-K / K-concatenated / grouped
GEMMs, the fused separator + PIT kernels with the separator-term recompute, `fast_backward`
accumulation into the flat bucket and the side-stream finalizers.  Every parameter gradient, the
loss, the SNR and the permutation indices of ONE such step are compared with the float64 torch-CPU
restatement (oracle/torch_ref.py) of main.py:208-337 + :354-358 (`compute_gradients` over all
trainables) on the same 32 mixtures.

Tolerance: gradients 2e-4 relative to the tensor's max (fp32 path vs float64 authority), loss and
SNR 1e-4 relative, permutation indices exact.
'''
import os
import time

import numpy as np
import pytest
import torch

from oracle import torch_ref as R
from test_gpu_fullsize import _setup, _synth, _cfg, relerr

pytestmark = pytest.mark.gpu
GTOL = 2e-4


@pytest.fixture(autouse=True)
def _lstm_status():
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


def _oracle_step(src, params, cfg):
    tp = {k: torch.tensor(v, dtype=torch.float64, requires_grad=True) for k, v in params.items()}
    r = R.model_forward(src.cpu().to(torch.complex128), tp, cfg)
    r['loss'].backward()
    return r, tp


def _train_step_vs_oracle(hp, model, src, min_checked):
    from danet_amd import ops
    model.keep_grads = True                  # the optimiser leaves the bucket readable
    assert model.fuse_heads                  # the path bench.py times
    params = model.param_dict()              # BEFORE the step (Adam moves them)
    out = model.train_step(src)
    torch.cuda.synchronize()
    assert ops.lstm_status_ok()
    t0 = time.time()
    ref, tp = _oracle_step(src, params, _cfg(hp))
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
    out, ref = _train_step_vs_oracle(hp, model, src, min_checked=14)
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
    out, ref = _train_step_vs_oracle(hp, model, src, min_checked=17)


def test_cfg4_h600_b32_train_step_gradients_vs_oracle(hp):
    '''BASELINE configs[3] as written (4 x 600): hoisted input GEMM + persistent forward, BPTT at
    H = 600, weight-gradient groups serial on the main stream.  Every layer's gradients (bottom
    and top included) against the oracle.'''
    model = _setup(hp, BATCH_SIZE=32, MAX_N_SIGNAL=3, EMBED_SIZE=40, NUM_LSTM_LAYERS=4,
                   LSTM_HDIM=600, TRAIN_ESTIMATOR_METHOD='truth-weighted')
    src = _synth(hp, 32, 128, 6)
    out, ref = _train_step_vs_oracle(hp, model, src, min_checked=17)


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


# ------------------------------------------------ data parallel, 2 ranks, the HIP path (one GPU)
_DP2_WORKER = r'''
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
         collectives=model.collectives_per_step(), status_tail=model._grad_store[-4:].cpu().numpy(),
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


def test_data_parallel_two_ranks_on_the_hip_path(hp, tmp_path):
    '''Model under torch.distributed with world_size 2 on the REAL kernels (both ranks on this
    box's one GPU, gloo moving the device buffers: RCCL cannot put two ranks on one device):
    shard by batch -> ONE all-reduce of the flat bucket -> 1/world inside the optimiser kernel.
    The summed bucket / 2 equals the gradient of ONE process on the global batch (the loss is a
    batch mean, app/ops.py:430), the step's parameters agree, the hand-off status words rode
    the bucket, and the replicas stay bit-identical over further steps.'''
    import subprocess
    import sys
    from danet_amd.model import Model
    hpd = dict(BATCH_SIZE=4, MAX_N_SIGNAL=2, FFT_SIZE=64, FFT_STRIDE=16, EMBED_SIZE=8, NUM_LSTM_LAYERS=2,
               LSTM_HDIM=32, NUM_ANCHOR=4, ENCODER_TYPE='bilstm-orig', TRAIN_ESTIMATOR_METHOD='anchor',
               INFER_ESTIMATOR_METHOD='anchor', SEPARATOR_TYPE='dot-softmax-orig')
    rng = np.random.RandomState(21)
    src = ((rng.randn(8, 2, 24, 33) + 1j * rng.randn(8, 2, 24, 33)) * 4).astype(np.complex64)
    np.save(tmp_path / 'src.npy', src)
    script = tmp_path / 'dp2_worker.py'
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script.write_text(_DP2_WORKER % dict(root=ROOT, hp=hpd))
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', DP_OUT=str(tmp_path))
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT'):
        env.pop(k, None)
    res = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                          '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)],
                         capture_output=True, text=True, timeout=600, env=env)
    assert res.returncode == 0, res.stdout[-1500:] + res.stderr[-3000:]
    r = [np.load(tmp_path / ('rank%d.npz' % i)) for i in range(2)]
    # the injected timeout: both ranks raised, at the same step (fault step + MAX_STEPS_IN_FLIGHT)
    from danet_amd import ops
    raised = [(tmp_path / ('fault%d.txt' % i)).read_text() for i in range(2)]
    assert raised[0] == raised[1] == str(2 + ops.MAX_STEPS_IN_FLIGHT), raised
    # one process, the global batch of 8, same initial parameters
    hp.load(dict(hpd, BATCH_SIZE=8))
    hp.digest()
    model = Model('dp2', device='cuda', seed=11).build()
    model.keep_grads = True
    p0 = model.param_dict()
    out = model.train_step(torch.as_tensor(src).cuda())
    torch.cuda.synchronize()
    g, p1 = model.grad_dict(), model.param_dict()
    assert int(r[0]['collectives']) == 1 and not r[0]['status_tail'].any() and not r[1]['status_tail'].any()
    assert abs(0.5 * (float(r[0]['loss']) + float(r[1]['loss'])) - float(out['loss'])) < 1e-5 * abs(float(out['loss']))
    for k in p0:
        assert np.array_equal(r[0]['p0:' + k], p0[k]) and np.array_equal(r[1]['p0:' + k], p0[k]), k
        assert np.array_equal(r[0]['g:' + k], r[1]['g:' + k]), k            # the same reduced bucket
        a, b = 0.5 * r[0]['g:' + k], g[k]
        assert np.abs(a - b).max() <= 2e-5 * (np.abs(b).max() + 1e-30), ('grad', k)
        assert np.array_equal(r[0]['p1:' + k], r[1]['p1:' + k]), k
        # Adam's first step is lr * sign-like: compare where the gradient is not at rounding level
        big = np.abs(b) > 1e-3 * np.abs(b).max()
        assert np.abs(r[0]['p1:' + k] - p1[k])[big].max() <= 1e-6 + 1e-4 * float(hp.LR), ('param', k)


def test_bench_two_ranks_functional(tmp_path):
    '''`python bench.py --gpus 2` end to end on this 1-GPU box (DANET_BENCH_TEST_SHARED_GPU=1: both
    ranks on device 0 over gloo -- a FUNCTIONAL run of the N > 1 path the driver takes on a
    multi-GPU node, not a measurement): the launcher spawns two ranks, both time the same K steps,
    rank 0 prints one JSON line with the whole-job value (2 x per-rank units / max-over-ranks time),
    one collective per step, weak scaling, and the e2e loop runs on every rank'''
    import json
    import subprocess
    import sys
    ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DANET_BENCH_TEST_SHARED_GPU='1')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_PORT', 'DANET_FORCE_DIST'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '4',
                          '--warmup', '1', '--no-cpu-baseline', '--batch', '8', '--frames', '32'],
                         capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res['n_gpus'] == 2 and res['rccl_ranks'] == 2 and res['scaling'] == 'weak'
    assert res['config']['global_batch'] == 16 and res['config']['parallelism'] == 'dp2'
    assert res['config']['collectives_per_step'] == 1 and res['allreduce_ms_standalone'] > 0
    per_step = 8 * 32 * 64 / 8000.0
    assert abs(res['value'] - 2 * per_step / (res['ms_per_step'] * 1e-3)) < 1e-2 * res['value']
    assert res['e2e']['ms_per_step'] > 0 and 'test_mode' in res
    # one run settles the gradient-reduction schedule: all three timed in the same
    # process group, next to the step without any reduction
    sc = res['schedules']
    assert set(sc['ms_per_step']) == {'0', 'tail', '1'} and all(v > 0 for v in sc['ms_per_step'].values())
    assert sc['no_reduction_ms_per_step'] > 0 and set(sc['exposed_comm_ms']) == {'0', 'tail', '1'}
    assert 'timed_region_note' in res
