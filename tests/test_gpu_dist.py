'''
GPU tests (run with -m gpu): data parallelism on the HIP path: reduction schedules, two ranks on one GPU, the bench spawn path.
Shared helpers: tests/gpu_helpers.py.
'''
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest
import torch
from gpu_helpers import FakeWork, ROOT, check_lstm_status, oracle_threads, rand_src, small_model

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _lstm_status():
    with oracle_threads():
        yield
    check_lstm_status()


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
        w = FakeWork(done)
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
        model = small_model(hp, BATCH_SIZE=32, FFT_SIZE=256, FFT_STRIDE=64, EMBED_SIZE=20,
                             NUM_LSTM_LAYERS=3, LSTM_HDIM=300, NUM_ANCHOR=6)
        model.keep_grads = True
        model.set_learn_rate(0.0)
        src = torch.as_tensor(rand_src(hp, 48, 1)).cuda()
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
    model = small_model(hp, BATCH_SIZE=32, FFT_SIZE=256, FFT_STRIDE=64, EMBED_SIZE=20,
                         NUM_LSTM_LAYERS=3, LSTM_HDIM=300, NUM_ANCHOR=6)
    model.keep_grads = True
    model.set_learn_rate(0.0)
    src = torch.as_tensor(rand_src(hp, 48, 1)).cuda()
    for _ in range(2):
        model.train_step(src)
    g1 = model.grad_dict()
    for k in g1:
        assert np.array_equal(2.0 * g1[k], grads['0'][k]), k


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
    # the schedule is a DECISION since round 6 (Model(grad_schedule='auto'), dist.choose_schedule): the
    # line says what was measured and what was picked, and the collective count follows the pick (gloo
    # staging a device bucket through the host is slow against this tiny step: 'tail' here)
    dec = res['config']['grad_allreduce_decision']
    assert dec is not None and dec['schedule'] == res['config']['grad_allreduce_schedule'] in ('0', 'tail')
    assert dec['allreduce_ms'] > 0 and dec['step_ms'] > 0 and dec['world'] == 2
    assert dec['schedule'] == ('tail' if dec['allreduce_ms'] > dec['threshold'] * dec['step_ms'] else '0')
    assert res['config']['collectives_per_step'] == {'0': 1, 'tail': 2}[dec['schedule']]
    assert res['allreduce_ms_standalone'] > 0
    per_step = 8 * 32 * 64 / 8000.0
    assert abs(res['value'] - 2 * per_step / (res['ms_per_step'] * 1e-3)) < 1e-2 * res['value']
    assert res['e2e']['ms_per_step'] > 0 and 'test_mode' in res
    # one run settles the gradient-reduction schedule: all three timed in the same
    # process group, next to the step without any reduction
    sc = res['schedules']
    assert set(sc['ms_per_step']) == {'0', 'tail', '1'} and all(v > 0 for v in sc['ms_per_step'].values())
    assert sc['no_reduction_ms_per_step'] > 0 and set(sc['exposed_comm_ms']) == {'0', 'tail', '1'}
    assert 'timed_region_note' in res
