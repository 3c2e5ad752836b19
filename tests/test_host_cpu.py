'''
CPU tests (no GPU) of the host logic around the hot path -- SURVEY 8(f-2):
the product's `utils.random_zeropad` against the reference golden G4, the
learning-rate schedule and NaN-restore of the train loop (main.py:439-476), and
the data-parallel pieces of that loop and of the default gradient-reduction
schedule (dist.TailOverlap) under gloo, world_size 2.
'''
import io
import os
import random
import subprocess
import sys
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, 'tests', 'golden', 'frontend_ref.npz')


def test_product_random_zeropad_matches_reference_golden():
    '''G4: danet_amd.utils.random_zeropad (the function the datasets call) under
    random.seed(k) == outputs of the reference's app/utils.py:78-92'''
    from danet_amd import utils
    gold = np.load(GOLD)
    base = gold['zeropad_base']
    for k in range(4):
        random.seed(k)
        assert np.array_equal(utils.random_zeropad(base, 5, axis=0), gold['zeropad_seed%d_axis0' % k])
        random.seed(k)
        assert np.array_equal(utils.random_zeropad(base, 7, axis=-1), gold['zeropad_seed%d_axis-1' % k])
    assert utils.random_zeropad(base, 0, axis=0) is base         # app/utils.py:83-84


class _StubModel(object):
    '''the surface cli.train touches, on CPU'''
    def __init__(self, losses):
        self.name = 'stub'
        self.device = torch.device('cpu')
        self.lr = None
        self.losses = list(losses)          # one per train_step
        self.i = 0
        self._flat = torch.zeros(3)
        self.saved, self.loaded, self.lr_log = [], [], []

    def set_learn_rate(self, lr):
        self.lr = float(lr)
        self.lr_log.append(self.lr)

    def get_learn_rate(self):
        return self.lr

    def train_step(self, spectra):
        v = self.losses[min(self.i, len(self.losses) - 1)]
        self.i += 1
        self._flat += 1
        return dict(loss=v, SNR=1.0, LR=self.lr)

    def valid_step(self, spectra):
        return dict(loss=0.5, SNR=1.0)

    def reset_state(self):
        pass

    def check_status(self):
        pass

    def save_params(self, fn, step=None):
        self.saved.append((fn, self._flat.clone()))

    def load_params(self, fn):
        self.loaded.append(fn)
        for f, t in self.saved:
            if f == fn:
                self._flat.copy_(t)
        return True


def _args(**kw):
    a = types.SimpleNamespace(no_save_on_epoch=False, no_valid_on_epoch=True)
    a.__dict__.update(kw)
    return a


def _toy_dataset(hp):
    hp.load(dict(DATASET_TYPE='toy', BATCH_SIZE=2, MAX_N_SIGNAL=2, FFT_SIZE=8, FFT_STRIDE=2,
                 MAX_TRAIN_LEN=16))
    hp.digest()
    ds = hp.get_dataset()()
    ds.install_and_load()
    return ds


@pytest.mark.parametrize('mode', ['fixed', 'adaptive', None])
def test_lr_decay_modes(hp, mode):
    '''main.py:439-459: 'fixed' decays every NUM_EPOCH_PER_LR_DECAY epochs, 'adaptive' only
    after that many epochs WITHOUT a new best loss, None never'''
    from danet_amd import cli
    hp.load(dict(LR=1.0, LR_DECAY=0.5, LR_DECAY_TYPE=mode, NUM_EPOCH_PER_LR_DECAY=2))
    m = _StubModel([])
    m.set_learn_rate(1.0)
    best, btime = float('inf'), 0
    lrs = []
    #           new best, new best, worse, worse -> decay, better, worse, worse -> decay
    for loss in [5.0, 4.0, 4.5, 4.2, 3.0, 3.5, 3.1]:
        best, btime = cli.lr_decay_update(m, loss, best, btime, io.StringIO())
        lrs.append(m.get_learn_rate())
    if mode == 'fixed':
        assert lrs == [1.0, 0.5, 0.5, 0.25, 0.25, 0.125, 0.125]
    elif mode == 'adaptive':
        assert lrs == [1.0, 1.0, 1.0, 0.5, 0.5, 0.5, 0.25]
        assert best == 3.0
    else:
        assert lrs == [1.0] * 7
    hp.LR_DECAY_TYPE = 'bogus'
    with pytest.raises(ValueError):
        cli.lr_decay_update(m, 1.0, best, btime, io.StringIO())


def test_nan_restore_redoes_the_epoch(hp, tmp_path, monkeypatch):
    '''main.py:462-476: an epoch whose mean metrics contain NaN restores the last
    checkpoint and is run again (the epoch counter does not advance); NaN in the very first
    epoch exits'''
    from danet_amd import cli
    monkeypatch.chdir(tmp_path)
    ds = _toy_dataset(hp)
    hp.load(dict(LR=1e-3, LR_DECAY_TYPE=None))
    nb = 10                                    # toy: 10 batches per epoch
    losses = [1.0] * nb + [1.0] * (nb - 1) + [float('nan')] + [0.5] * (2 * nb)
    m = _StubModel(losses)
    out = io.StringIO()
    cli.train(m, 3, ds, _args(), out)
    # epoch 1 ok (saved _e1), epoch 2 NaN -> restore _e1, redo epoch 2, epoch 3
    assert m.loaded == ['saves/stub_e1']
    assert [f for f, _ in m.saved] == ['saves/stub_e1', 'saves/stub_e2', 'saves/stub_e3']
    assert m.i == 4 * nb and 'restoring last checkpoint' in out.getvalue()
    # parameters after the restore were those saved at the end of epoch 1 (+ the redone epochs)
    assert float(m.saved[1][1][0]) == float(m.saved[0][1][0]) + nb
    m2 = _StubModel([float('nan')])
    with pytest.raises(SystemExit):
        cli.train(m2, 2, ds, _args(), io.StringIO())


_DP_LOOP_WORKER = r'''
import io, os, sys, types
sys.path.insert(0, %(root)r)
sys.path.insert(0, os.path.join(%(root)r, 'tests'))
import numpy as np, torch
import __graft_entry__ as g; g.load_package()
from danet_amd import dist, cli
from danet_amd.hparams import hparams
import test_host_cpu as T
dist.init_from_env('gloo')
rank, world = dist.rank(), dist.world_size()
assert world == 2
os.chdir(os.environ['DP_OUT'])
ok = True
# --- epoch metrics are averaged over ranks; NaN on ONE rank is seen by both
vals = dist.allreduce_mean_scalars([1.0 + rank, float('nan') if rank == 1 else 2.0], 'cpu')
ok = ok and vals[0] == 1.5 and vals[1] != vals[1]
# --- the whole loop: rank 1 alone hits NaN in epoch 2; BOTH ranks must restore and redo it,
#     and end with identical parameters (rank 0 reads the file, the others get a broadcast)
hparams.reset()
ds = T._toy_dataset(hparams)
hparams.load(dict(LR=1.0, LR_DECAY=0.5, LR_DECAY_TYPE='adaptive', NUM_EPOCH_PER_LR_DECAY=1))
nb = 10
if rank == 1:
    losses = [1.0] * nb + [1.0] * (nb - 1) + [float('nan')] + [0.5] * (2 * nb)
else:
    losses = [1.0] * (2 * nb) + [0.5] * (2 * nb)
m = T._StubModel(losses)
m._flat += 100.0 * rank                     # replicas that would drift apart without the broadcast
out = io.StringIO()
cli.train(m, 3, ds, T._args(), out)
ok = ok and m.i == 4 * nb and 'restoring last checkpoint' in out.getvalue()
ok = ok and (rank != 0 or m.loaded == ['saves/stub_e1']) and (rank != 1 or m.loaded == [])
ok = ok and (rank != 1 or m.saved == [])     # only rank 0 writes files
# after the restore both ranks hold rank 0's checkpointed parameters + the same number of steps
t = m._flat.clone()
torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
ok = ok and bool(torch.equal(t, m._flat)) and float(m._flat[0]) == 4 * nb - nb
# ADVICE r4: the restore's broadcast reaches the other ranks through a c10d collective, which does
# not bump torch's version counter -- dist.broadcast_params_ itself must mark the packed weights
# that overlap the received range stale (rank 1 would otherwise keep its pre-restore packs)
from danet_amd import ops
w = torch.zeros(64) + rank
pw = object.__new__(ops._PackedWeight)
pw.W, pw.lo, pw.hi, pw.stale, pw.version, pw.unused = w, w.data_ptr() + 16, w.data_ptr() + 64, False, w._version, 0
other = torch.zeros(8)
po = object.__new__(ops._PackedWeight)
po.W, po.lo, po.hi, po.stale, po.version, po.unused = other, other.data_ptr(), other.data_ptr() + 32, False, 0, 0
ops._packs[ops._dev_key(w.device)] = {'w': pw, 'other': po}
v0 = w._version
dist.broadcast_params_(w)
ok = ok and float(w[5]) == 0.0 and pw.stale and not po.stale and (rank == 0 or w._version == v0 or pw.stale)
ops._packs.clear()
# identical learning-rate history on both ranks (decisions taken on the rank-mean loss)
lr = torch.tensor(m.lr_log + [0.0] * (8 - len(m.lr_log)))
lr2 = lr.clone(); torch.distributed.all_reduce(lr2, op=torch.distributed.ReduceOp.MAX)
ok = ok and bool(torch.equal(lr, lr2))
# --- default gradient-reduction schedule: 'rest' piece from the hook + bottom layer in finish()
sizes = [7, 40, 24, 13, 5]
flat_p = torch.zeros(sum(sizes))
flat_g = torch.arange(sum(sizes), dtype=torch.float32) * (rank + 1) + 0.25 * rank
expect = torch.arange(sum(sizes), dtype=torch.float32) * 3 + 0.25
views, offs, off = [], {}, 0
for n in sizes:
    v = flat_p[off:off + n]
    views.append(v); offs[v.data_ptr()] = (off, off + n); off += n
for bottom in ([views[0], views[1]], [views[2]], [views[4]]):
    for trial in range(2):
        gbuf = flat_g.clone()
        b = dist.TailOverlap(gbuf, offs)
        b.hook(('layer', 1), [views[3]])          # per-layer events are not this schedule's
        assert not b.works
        b.hook(('rest',), bottom)
        n_pieces = len(b.works)
        b.hook(('rest',), bottom)                 # fires once per step
        assert len(b.works) == n_pieces and n_pieces in (1, 2)
        scale = b.finish()
        ok = ok and scale == 0.5 and bool(torch.equal(gbuf, expect)) and not b.works and not b.fired
# no 'rest' event at all (e.g. the toy encoder): finish() reduces the whole bucket
gbuf = flat_g.clone()
b = dist.TailOverlap(gbuf, offs)
ok = ok and b.finish() == 0.5 and bool(torch.equal(gbuf, expect))
open(os.path.join(os.environ['DP_OUT'], 'loop' + str(rank) + '.txt'), 'w').write('OK' if ok else 'BAD')
'''


def test_data_parallel_train_loop_and_tail_overlap_gloo_world2(tmp_path):
    '''rank-collective LR / NaN-restore decisions (cli.train) and dist.TailOverlap == one
    all-reduce of the whole bucket, under gloo with 2 processes'''
    script = tmp_path / 'loop_worker.py'
    script.write_text(_DP_LOOP_WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', DP_OUT=str(tmp_path))
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    out = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
         '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)],
        capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    for r in range(2):
        assert (tmp_path / ('loop%d.txt' % r)).read_text() == 'OK'


def test_bench_cli_surface():
    '''bench.py: configs named by BASELINE.json exist, the spawn path is taken exactly when no
    torchrun environment is present and N > 1 (or DANET_FORCE_DIST=1)'''
    sys.path.insert(0, ROOT)
    import bench
    assert {'cfg2', 'cfg4', 'cfg4h600', 'cfg5', 'cfg5-kmeans'} <= set(bench.CONFIGS)
    c = bench.CONFIGS
    assert (c['cfg2']['batch'], c['cfg2']['layers'], c['cfg2']['hdim']) == (32, 3, 300)
    assert c['cfg4h600']['hdim'] == 600 and c['cfg4']['hp']['MAX_N_SIGNAL'] == 3 \
        and c['cfg4']['hp']['EMBED_SIZE'] == 40
    assert c['cfg5']['kind'] == 'infer' and c['cfg5']['hp']['FFT_SIZE'] == 512 \
        and c['cfg5']['frames'] == 1251
    a = types.SimpleNamespace(gpus=1)
    env = dict(os.environ)
    try:
        os.environ.pop('WORLD_SIZE', None)
        os.environ.pop('DANET_FORCE_DIST', None)
        assert bench.maybe_spawn(a) is None                  # N = 1: runs in place
        os.environ['WORLD_SIZE'] = '4'
        assert bench.maybe_spawn(types.SimpleNamespace(gpus=4)) is None   # already under torchrun
        os.environ.pop('WORLD_SIZE')
        with pytest.raises(SystemExit) as e:                 # N > visible GPUs: clear error
            bench.maybe_spawn(types.SimpleNamespace(gpus=64))
        assert 'GPU(s) visible' in str(e.value)
    finally:
        os.environ.clear()
        os.environ.update(env)


# ------------------------------------------------------------------ round 4: the feed
def _varlen_batches(hp, n, seed=3):
    '''host batches [B*C, T_i, F] of different lengths, some real some complex (toy data is
    real, main.py:418-421)'''
    rng = np.random.RandomState(seed)
    out = []
    for i in range(n):
        T = int(rng.randint(10, 40))
        a = rng.rand(hp.BATCH_SIZE * hp.MAX_N_SIGNAL, T, hp.FEATURE_SIZE).astype(np.float32)
        if i % 2:
            a = (a + 1j * rng.rand(*a.shape)).astype(np.complex64)
        out.append((a,))
    return out


def test_batch_feed_ahead_equals_synchronous_loop(hp):
    '''feed.BatchFeed: the feed yields exactly the tensors of the reference's synchronous loop
    (reshape, cast to complex64, random crop to MAX_TRAIN_LEN with the SAME draws from python's
    `random`, main.py:417-426); batch i+1 has been fetched (its upload issued) before the consumer
    gets batch i, batch i+2 is fetched only after the consumer is back'''
    from danet_amd import feed
    _toy_dataset(hp)
    batches = _varlen_batches(hp, 9)
    random.seed(11)
    sync = [t.clone() for t in feed.BatchFeed(iter(batches), 'cpu', hp.MAX_TRAIN_LEN, mode='sync')]
    random.seed(11)
    pulled = []

    def source():
        for i, b in enumerate(batches):
            pulled.append(i)
            yield b
    ahead = []
    for i, t in enumerate(feed.BatchFeed(source(), 'cpu', hp.MAX_TRAIN_LEN, mode='ahead')):
        assert pulled[-1] == min(i + 1, 8)
        ahead.append(t.clone())
    assert len(sync) == len(ahead) == 9
    for a, b, (raw,) in zip(sync, ahead, batches):
        assert a.dtype == b.dtype == torch.complex64 and torch.equal(a, b)
        assert a.shape == (hp.BATCH_SIZE, hp.MAX_N_SIGNAL, min(raw.shape[1], hp.MAX_TRAIN_LEN),
                           hp.FEATURE_SIZE)
    # the reference's crop by hand (main.py:422-426) on the same random stream
    random.seed(11)
    for a, (raw,) in zip(sync, batches):
        x = raw.reshape(hp.BATCH_SIZE, hp.MAX_N_SIGNAL, -1, hp.FEATURE_SIZE)
        if x.shape[2] > hp.MAX_TRAIN_LEN:
            beg = random.randint(0, x.shape[2] - hp.MAX_TRAIN_LEN - 1)
            x = x[:, :, beg:beg + hp.MAX_TRAIN_LEN]
        assert np.array_equal(a.numpy(), x.astype(np.complex64))


def test_batch_feed_propagates_dataset_errors(hp):
    from danet_amd import feed
    _toy_dataset(hp)

    def bad():
        yield _varlen_batches(hp, 1)[0]
        raise RuntimeError('corpus file missing')
    it = iter(feed.BatchFeed(bad(), 'cpu', None, mode='ahead'))
    next(it)
    with pytest.raises(RuntimeError, match='corpus file missing'):
        next(it)
    assert list(feed.BatchFeed(iter([]), 'cpu', None, mode='ahead')) == []


def test_step_report_equals_running_float_sum():
    '''feed.StepReport == the reference's `dst[k] += float(v)` / `* 1/(n)` (main.py:433-436)
    bit for bit, for device-style scalars and python floats, across a flush boundary'''
    from danet_amd import feed
    rng = np.random.RandomState(0)
    vals = rng.randn(2500).astype(np.float32) * 1e3
    rep = feed.StepReport(flush_every=1024)
    ref = {}
    for i, v in enumerate(vals):
        fetch = dict(loss=torch.tensor(v), SNR=torch.tensor(np.float32(v * 0.5)), LR=3e-4)
        rep.add(fetch)
        for k, x in fetch.items():
            ref[k] = ref.get(k, 0.) + float(x)
    got = rep.mean()
    assert list(got) == ['loss', 'SNR', 'LR']
    for k in ref:
        assert got[k] == ref[k] * (1. / len(vals)), k
    nan = feed.StepReport()
    nan.add(dict(loss=torch.tensor(float('nan'))))
    assert nan.mean()['loss'] != nan.mean()['loss']


_DP_FAULT_WORKER = r'''
import collections, datetime, os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
import __graft_entry__ as g; g.load_package()
from danet_amd import ops, _lib
torch.distributed.init_process_group('gloo', timeout=datetime.timedelta(seconds=60))
rank = torch.distributed.get_rank()
assert torch.distributed.get_world_size() == 2
# a CPU stand-in for the device's status record (ops._DeviceStatus needs pinned memory): the word
# is the 4-float tail of a gradient bucket, as Model sets it up under data parallelism
n = 10
grad_store = torch.zeros(n + 4)
st = object.__new__(ops._DeviceStatus)
st.dev, st.own = torch.device('cuda', 0), torch.zeros(4, dtype=torch.int32)
st.word, st.host_mapped = grad_store[n:].view(torch.int32), False
st.slots = torch.zeros(ops.MAX_STEPS_IN_FLIGHT + 2, 4, dtype=torch.int32)
st.queue, st.n, st.retired, st.events = collections.deque(), 0, 0, []
ops._status[0] = st
torch.cuda.synchronize = lambda *a, **k: None
rng = np.random.RandomState(100 + rank)       # how far "this rank's GPU" happens to be: differs per rank

class FakeEvent(object):
    def query(self):
        return bool(rng.rand() < 0.5)
    def synchronize(self):
        pass
ops._record_event = lambda st=None: FakeEvent()
FAULT_STEP = 6
raised_at, steps_enqueued = None, 0
for step in range(20):
    try:
        ops.poll_status(st.dev)                # admission (Model.train_step's first line)
    except _lib.DanetHipError:
        raised_at = step
        break
    grad_store[:n] = float(step)               # "backward"
    if rank == 1 and step == FAULT_STEP:
        grad_store[n] = 1.0                    # the persistent kernel's DANET_STATUS_TIMEOUT store
    torch.distributed.all_reduce(grad_store)   # the step's ONE gradient all-reduce carries the word
    ops.step_done(st.dev)
    steps_enqueued += 1
open(os.path.join(os.environ['DP_OUT'], 'fault%%d.txt' %% rank), 'w').write(
    '%%s %%d' %% (raised_at, steps_enqueued))
'''


def test_handoff_timeout_raises_at_the_same_step_on_every_rank_gloo_world2(tmp_path):
    '''ADVICE r3 (ops.poll_status): a hand-off timeout on ONE rank is seen by all ranks through
    the gradient all-reduce, and every rank raises DanetHipError at the admission of the SAME
    step (fault step + MAX_STEPS_IN_FLIGHT) with the same number of collectives enqueued --
    whatever the completion state of each rank's events'''
    from danet_amd import ops
    script = tmp_path / 'fault_worker.py'
    script.write_text(_DP_FAULT_WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', DP_OUT=str(tmp_path))
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    out = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
         '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)],
        capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-3000:]
    res = [(tmp_path / ('fault%d.txt' % r)).read_text() for r in range(2)]
    expect = '%d %d' % (6 + ops.MAX_STEPS_IN_FLIGHT, 6 + ops.MAX_STEPS_IN_FLIGHT)
    assert res == [expect, expect], res
