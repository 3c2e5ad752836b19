'''
CPU tests (run with -m "not gpu") of the drop-in boundary: the C-ABI library
loads and exports every symbol include/danet_hip.h declares (no compute calls
without a GPU), the hparams / registry surface matches the reference's
(app/hparams.py, default.json), the product path fails loudly instead of
falling back, and the N>1 data-parallel path (gloo, world_size 2).
'''
import ctypes
import json
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    txt = open(os.path.join(ROOT, 'include', 'danet_hip.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(danet_[a-z0-9_]+)\s*\(', txt)))


def test_library_loads_and_exports_every_declared_symbol():
    from danet_amd import _lib
    lib = _lib.load()                       # dlopen + resolves every PROTOTYPES entry
    syms = _header_symbols()
    assert len(syms) >= 30
    for s in syms:
        assert hasattr(lib, s), 'missing export: ' + s
    assert set(_lib.PROTOTYPES) == set(syms), set(_lib.PROTOTYPES) ^ set(syms)
    assert lib.danet_abi_version() == 7
    assert len(syms) <= 51                  # round 4: the ABI is what ships, not every experiment
    # ... and NOTHING else of the library's own is exported (-fvisibility=hidden + csrc/exports.map):
    # the dynamic symbol table holds the declared entry points only
    import subprocess
    out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIB_PATH], capture_output=True, text=True, check=True)
    exported = sorted(l.split()[-1] for l in out.stdout.splitlines() if l.strip())
    assert exported and all(e.startswith('danet_') for e in exported), [e for e in exported if not e.startswith('danet_')][:5]
    assert set(exported) == set(syms), set(exported) ^ set(syms)
    # pure host-side helpers are callable without a GPU
    assert lib.danet_stft_num_frames(8000, 256, 64) == 126
    assert lib.danet_stft_num_frames(160000, 512, 128) == 1251
    assert lib.danet_stft_num_frames(255, 256, 64) < 0
    # the ONE scratch-size query (danet_workspace_bytes): every DANET_WS_* op answers, a wrong
    # op / dim count is an error value, not a crash
    assert _lib.ws_bytes(_lib.WS_CENTER_MEAN, 4) >= 16
    assert _lib.ws_bytes(_lib.WS_GEMM, 300, 1200, 4096) > 0
    assert _lib.ws_bytes(_lib.WS_GEMM, 4096, 1200, 600) == 0
    dims = {_lib.WS_ISTFT: (2, 128, 256, 64), _lib.WS_GEMM_STREAMK: (128, 128, 128),
            _lib.WS_COLSUM: (4096, 1200), _lib.WS_LSTM: (128, 32, 300, 2),
            _lib.WS_ATTRACTOR_TRUTH: (32, 2, 16512, 20), _lib.WS_ATTRACTOR_ANCHOR: (32, 2, 16512, 20, 6),
            _lib.WS_SEPARATE_BWD: (32, 2, 16512, 20), _lib.WS_SEPARATE_PIT: (32, 2, 16512, 20),
            _lib.WS_SEPARATE_PIT_RECORDS: (32, 16512), _lib.WS_PIT_MSE: (32, 2, 16512)}
    for op, d in dims.items():
        assert 0 < _lib.ws_bytes(op, *d) < (1 << 32), op
    txt = open(os.path.join(ROOT, 'include', 'danet_hip.h')).read()
    assert re.search(r'DANET_WS_GEMM_X6_TN,[^\n]*\n\s*DANET_WS_SEPARATE_PIT_GRAD,.{0,300}?DANET_WS_COUNT', txt, re.S)   # enum order == _lib's
    assert _lib.WS_CENTER_MEAN == 11 and _lib.WS_GEMM_PACK == 13 and _lib.WS_GEMM_X6_TN == 14
    assert _lib.WS_SEPARATE_PIT_GRAD == 15
    # the attractor-gradient partials of the fused separator + loss forward: C! x C x EP sums per chunk,
    # offered for C == 2 (0 = not offered: the caller keeps danet_separate_pit_bwd)
    assert _lib.ws_bytes(_lib.WS_SEPARATE_PIT_GRAD, 32, 2, 16512, 20) == 32 * 9 * 2 * 2 * 20 * 4
    assert _lib.ws_bytes(_lib.WS_SEPARATE_PIT_GRAD, 32, 3, 16512, 40) == 0 == _lib.ws_bytes(_lib.WS_SEPARATE_PIT_GRAD, 32, 1, 16512, 20)
    # a BiLSTM layer's four weight-gradient products at cfg 2: 160 tiles -> 3 K slices
    assert _lib.ws_bytes(_lib.WS_GEMM_X6_TN, 2 * 900 * 1200, 160, 4096) == 3 * 2 * 900 * 1200 * 4
    assert _lib.ws_bytes(_lib.WS_GEMM_X6, 4096, 2580, 600, 0) == 0                # enough tiles: no K slices
    # dX of a BiLSTM layer: 160 tiles -> 3 K slices: 16 KB of tickets + the slabs
    assert _lib.ws_bytes(_lib.WS_GEMM_X6, 4096, 600, 1200, 1200) == 16384 + 3 * 4096 * 600 * 4
    # three bf16 pieces, 128-column panels, 16-k steps: 600 -> 640 columns, 2580 -> 162 steps
    assert _lib.ws_bytes(_lib.WS_GEMM_PACK, 600, 2580) == 3 * 640 * 162 * 16 * 2
    arr = (ctypes.c_int64 * 3)(1, 2, 3)
    assert lib.danet_workspace_bytes(_lib.WS_LSTM, arr, 3) == ctypes.c_size_t(-1).value
    assert lib.danet_workspace_bytes(99, arr, 3) == ctypes.c_size_t(-1).value
    with pytest.raises(_lib.DanetHipError):
        _lib.ws_bytes(_lib.WS_LSTM, 1, 2, 3)


def test_abi_has_no_torch_types():
    txt = open(os.path.join(ROOT, 'include', 'danet_hip.h')).read()
    assert 'extern "C"' in txt
    code = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)          # declarations only
    assert 'torch' not in code.lower() and 'at::' not in code and 'Tensor' not in code
    assert '#include <torch' not in txt and '#include <ATen' not in txt


def test_bad_arguments_return_error_codes_not_crashes():
    from danet_amd import _lib
    lib = _lib.load()
    rc = lib.danet_gemm_f32(None, 0, 0, 0, 4, 4, None, 4, None, 4, None, 4, None, 0.0, None, 0, 0)
    assert rc == -1 and b'gemm' in lib.danet_last_error()
    rc = lib.danet_lstm_fwd(None, 4, 2, 6, 2, None, None, None, None, 24, None, 12,
                            None, None, None, None, ctypes.c_void_p(8), 64, None, 0)
    assert rc == -3 and b'multiple of 4' in lib.danet_last_error()
    with pytest.raises(_lib.DanetHipError):
        _lib.check(rc)


def test_product_path_fails_loudly_without_gpu_or_extension(tmp_path):
    '''no CPU fallback: CPU tensors are rejected, and a missing .so is a hard error'''
    from danet_amd import ops, _lib
    with pytest.raises(AssertionError):
        ops.frontend(torch.zeros(1, 2, 3, 5, dtype=torch.complex64))       # CPU tensor
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import __graft_entry__ as g; g.load_package()\n"
        "from danet_amd import _lib\n"
        "_lib.LIB_PATH = %r\n"
        "try:\n"
        "    _lib.load()\n"
        "except _lib.DanetHipError as e:\n"
        "    print('LOUD:', 'no CPU fallback' in str(e))\n"
    ) % (ROOT, str(tmp_path / 'nope.so'))
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=300)
    assert 'LOUD: True' in out.stdout, out.stdout + out.stderr


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'danet-tensorflow_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(('.py', '.hip', '.h', '.cpp')):
                txt = open(os.path.join(dirpath, f)).read()
                for pat in (r'^\s*(from|import)\s+oracle', r'import_module\([\'"]oracle',
                            r'danet_oracle', r'torch_ref', r'oracle/'):
                    assert not re.search(pat, txt, flags=re.M), (os.path.join(dirpath, f), pat)


# ------------------------------------------------------------------ hparams
def test_hparams_surface_matches_reference_default_json(hp):
    # tests/golden/hparams_ref.{json,npz}: what the REFERENCE's own Hyperparameter.load_json + digest
    # (app/hparams.py:26-69) make of its default.json -- every key with its value, the derived
    # COMPLEXX / FEATURE_SIZE and the bit pattern of the evaluated FFT_WND (make_golden_hparams.py)
    gold = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'hparams_ref.json')))
    wnd = np.load(os.path.join(ROOT, 'tests', 'golden', 'hparams_ref.npz'))
    assert len(gold['loaded']) == 31
    for k, v in gold['loaded'].items():
        if k != 'FFT_WND':                                     # (ours keeps the expression, as the reference does until digest)
            assert getattr(hp, k) == v, k
    assert hp.NUM_LSTM_LAYERS == 4 and hp.LSTM_HDIM == 300      # reference hard-codes these
    for name in gold['registries'] + gold['accessors']:
        if name != 'get_regularizer':                          # (REG_* are dead in the reference: main.py never calls it)
            assert hasattr(hp, name), name
    hp.digest()
    assert hp.COMPLEXX == gold['derived']['COMPLEXX'] and hp.FEATURE_SIZE == gold['derived']['FEATURE_SIZE']
    assert str(hp.FFT_WND.dtype) == gold['derived']['FFT_WND_dtype']
    assert np.array_equal(hp.FFT_WND.view(np.uint32), wnd['FFT_WND_256'].view(np.uint32))     # bit for bit
    hp.load(dict(FFT_SIZE=512))
    hp.digest()                                                # re-derives from the expression
    assert hp.FEATURE_SIZE == gold['derived_fft512']['FEATURE_SIZE']
    assert np.array_equal(hp.FFT_WND.view(np.uint32), wnd['FFT_WND_512'].view(np.uint32))
    with pytest.raises(NameError):
        hp.load({'lower_case': 1})
    with pytest.raises(AssertionError):
        hp.load({'BAD': [1, 2]})


def test_hparams_load_json(hp, tmp_path):
    f = tmp_path / 'c.json'
    f.write_text(json.dumps(dict(BATCH_SIZE=8, ENCODER_TYPE='bilstm-orig')))
    hp.load_json(str(f))
    assert hp.BATCH_SIZE == 8 and hp.ENCODER_TYPE == 'bilstm-orig'


def test_plugin_registries_and_class_contracts(hp):
    from danet_amd import modules
    assert set(hp.encoder_registry) >= {'toy', 'lstm-orig', 'bilstm-orig'}
    assert set(hp.estimator_registry) >= {'truth', 'truth-threshold', 'truth-weighted', 'anchor'}
    assert set(hp.separator_registry) == {'dot-sigmoid-orig', 'dot-softmax-orig'}
    assert set(hp.ozer_registry) == {'sgd', 'adam'}
    assert 'toy' in hp.dataset_registry
    hp.load(dict(ENCODER_TYPE='bilstm-orig'))
    assert hp.get_encoder() is modules.BiLstmEncoder
    assert hp.get_estimator('anchor') is modules.AnchoredEstimator
    assert hp.get_separator('dot-softmax-orig') is modules.DotSeparatorSoftmax
    assert modules.AnchoredEstimator.USE_TRUTH is False
    for n in ('truth', 'truth-threshold', 'truth-weighted'):
        assert hp.get_estimator(n).USE_TRUTH is True
    with pytest.raises(KeyError):
        hp.get_estimator('kmeans-not-registered')
    for cls in (modules.Encoder, modules.Estimator, modules.Separator):
        with pytest.raises(NotImplementedError):
            cls(None, 'x')(None) if cls is not modules.Separator else cls(None, 'x')(None, None, None)
    m = modules.BiLstmEncoder('model', 'encoder')
    assert m.model == 'model' and m.name == 'encoder'


def test_user_plugin_registration(hp):
    from danet_amd.hparams import hparams
    from danet_amd import modules

    @hparams.register_estimator('my-est')
    class MyEst(modules.Estimator):
        USE_TRUTH = False
    try:
        assert hp.get_estimator('my-est') is MyEst
    finally:
        del type(hp).estimator_registry['my-est']


def test_toy_dataset_matches_reference_shape(hp):
    from danet_amd import datasets
    hp.digest()
    ds = hp.get_dataset()()
    with pytest.raises(RuntimeError):
        next(ds.epoch('train', 8))
    ds.install_and_load()
    batches = list(ds.epoch('train', 8))
    assert len(batches) == 10
    sig, = batches[0]
    assert sig.shape == (8, 128, 129) and sig.dtype == np.float32
    assert 0.0 <= sig.min() and sig.max() < 1.0
    waves = datasets.synth_waves(1337, 3, 128)
    assert waves.shape == (3, 127 * 64) and waves.dtype == np.float32
    assert abs(np.sqrt(np.mean(waves[0] ** 2)) - 1000.0) < 1.0


def test_toy_dataset_stream_against_reference_golden(hp):
    '''the `toy` generator's STREAM (not just its shape) against arrays produced by the
    reference's own app/datasets/dataset.py:43-63 under np.random.seed(1337)
    (tests/golden/make_golden_toy.py -> toy_ref.npz): bit-exact, float32, 10 batches per
    epoch, the next epoch continues the same numpy stream, unloaded -> RuntimeError'''
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'toy_ref.npz'))
    hp.load(dict(DATASET_TYPE='toy', FFT_SIZE=8, FFT_STRIDE=2))
    hp.digest()
    assert hp.FEATURE_SIZE == 5
    ds = hp.get_dataset()()
    assert int(gold['unloaded_raises']) == 1
    with pytest.raises(RuntimeError):
        next(ds.epoch('train', 4))
    ds.install_and_load()
    np.random.seed(1337)
    ep = np.stack([b for (b,) in ds.epoch('train', 4, shuffle=True)])
    assert ep.dtype == np.float32 and np.array_equal(ep, gold['toy_f5_seed1337'])
    nxt, = next(ds.epoch('valid', 4))
    assert np.array_equal(nxt, gold['toy_f5_seed1337_second_epoch_first'])
    # BASELINE cfg 1's shape (F = 129, B*C = 8)
    hp.reset()
    hp.load(dict(DATASET_TYPE='toy'))
    hp.digest()
    ds = hp.get_dataset()()
    ds.install_and_load()
    np.random.seed(1337)
    ep = [b for (b,) in ds.epoch('train', 8)]
    assert len(ep) == int(gold['toy_f129_n_batches'])
    assert tuple(ep[0].shape) == tuple(gold['toy_f129_shape']) and str(ep[0].dtype) == str(gold['toy_f129_dtype'])
    assert np.array_equal(np.array([b.astype(np.float64).sum() for b in ep]), gold['toy_f129_sums'])
    assert np.array_equal(ep[0][:, 0, :], gold['toy_f129_first_rows'])
    assert np.array_equal(ep[-1][:, -1, :], gold['toy_f129_last_rows'])


# -------------------------------------------------- data parallel (gloo, N=2)
_DP_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch
import __graft_entry__ as g; g.load_package()
from danet_amd import dist
from oracle import danet_oracle as O, torch_ref as R
dist.init_from_env('gloo')
rank, world = dist.rank(), dist.world_size()
assert world == 2
rng = np.random.RandomState(0)
Bg, C, T, F, E, H, L, A = 4, 2, 6, 5, 3, 4, 1, 4
src = (rng.randn(Bg, C, T, F) + 1j * rng.randn(Bg, C, T, F)) * 3
p = O.init_bilstm_params(rng, F, E, H, L)
p['global/train_estimator/anchors'] = rng.randn(A, E)
cfg = dict(H=H, L=L, E=E, C=C, A=A, train_est='anchor', infer_est='anchor', separator='dot-softmax-orig')
names = sorted(p)
def flat_grad(batch, pd):
    tp = {k: torch.tensor(v, requires_grad=True) for k, v in pd.items()}
    R.model_forward(torch.tensor(batch), tp, cfg)['loss'].backward()
    return torch.cat([tp[k].grad.reshape(-1) for k in names])
# rank-dependent garbage params, then broadcast from rank 0
flat_p = torch.cat([torch.tensor(p[k]).reshape(-1) for k in names]) + (0.0 if rank == 0 else 7.0)
dist.broadcast_params_(flat_p)
off = 0
pd = {}
for k in names:
    n = p[k].size
    pd[k] = flat_p[off:off + n].reshape(p[k].shape).numpy().copy(); off += n
shard = src[rank * 2:(rank + 1) * 2]            # BATCH_SIZE = 2 per rank
gbuf = flat_grad(shard, pd)
scale = dist.allreduce_grads_(gbuf)              # ONE collective per step
gbuf *= scale
full = flat_grad(src, p)                         # global-batch gradient, rank-0 params
err = float((gbuf - full).abs().max() / full.abs().max())
mx = dist.allreduce_max_scalar(float(rank + 1), 'cpu')
open(os.path.join(os.environ['DP_OUT'], 'rank' + str(rank) + '.txt'), 'w').write(
    ' '.join(['RANK', str(rank), 'ERR', repr(err), 'MAX', repr(mx), 'SEED', str(dist.shard_seed(1337))]))
'''


def test_data_parallel_gloo_world2(tmp_path):
    '''sharding by batch + ONE summed all-reduce + 1/world == global-batch gradient'''
    script = tmp_path / 'dp_worker.py'
    script.write_text(_DP_WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', DP_OUT=str(tmp_path))
    import socket
    with socket.socket() as sk:                     # a free port (a fixed one lingers in TIME_WAIT)
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    out = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
         '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)],
        capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    lines = [(tmp_path / ('rank%d.txt' % r)).read_text() for r in range(2)]   # not stdout: it interleaves
    for l in lines:
        tok = l.split()
        assert float(tok[3]) < 1e-12, l
        assert float(tok[5]) == 2.0
        assert int(tok[7]) == 1337 + int(tok[1])


_BUCKET_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch
import __graft_entry__ as g; g.load_package()
from danet_amd import dist
dist.init_from_env('gloo')
rank, world = dist.rank(), dist.world_size()
assert world == 2
# a flat gradient bucket with five "parameters" (views), as Model._flatten lays them out
sizes = [7, 40, 24, 13, 5]
flat_p = torch.zeros(sum(sizes))
flat_g = torch.arange(sum(sizes), dtype=torch.float32) * (rank + 1) + 0.25 * rank
expect = torch.arange(sum(sizes), dtype=torch.float32) * 3 + 0.25          # rank 0 + rank 1
views, offs, off = [], {}, 0
for n in sizes:
    v = flat_p[off:off + n]
    views.append(v); offs[v.data_ptr()] = (off, off + n); off += n
ok = True
for trial in range(3):                       # the bucket object is reused step after step
    gbuf = flat_g.clone()
    b = dist.GradBuckets(gbuf, offs)
    b.hook(('out',), [views[4]])                         # last variable first (backward order)
    b.hook(('layer', 1), [views[2], views[3]])           # contiguous pair -> one collective
    b.hook(('layer', 0), [views[0], views[2]])           # NOT contiguous -> left to finish()
    b.hook(('x',), [torch.zeros(3)])                     # a tensor that is not ours -> ignored
    assert len(b.works) == 2 and sorted(b.covered) == [(47, 84), (84, 89)]
    scale = b.finish()                                   # reduces [0, 47) and waits for all
    ok = ok and scale == 0.5 and bool(torch.equal(gbuf, expect)) and not b.works and not b.covered
# 'tail' schedule over a store with the 4 status words IN FRONT (Model._flatten, round 6): the early piece
# leaves them alone, they ride the LAST collective -- behind every kernel of the step, wherever a schedule
# launches its early pieces
store = torch.cat([torch.zeros(4), flat_g.clone()])
store[0] = float(rank)                                   # "rank 1 timed out"
offs4 = {k: (lo + 4, hi + 4) for k, (lo, hi) in offs.items()}
t = dist.TailOverlap(store, offs4, last=[(0, 4)])
t.hook(('rest',), [views[0], views[1]])                  # bottom layer = the first two variables
assert t.launched == 1 and t.covered == [(4 + 47, 4 + 89)], t.covered
t.wait_launched()
status_after_early_piece = float(store[0])
scale = t.finish()
ok = ok and scale == 0.5 and status_after_early_piece == float(rank) and float(store[0]) == 1.0
ok = ok and bool(torch.equal(store[4:], expect))
# the measured decision: every rank gets the same (MAX-reduced) time, hence the same schedule
ms = dist.measure_allreduce_ms(1 << 16, torch.device('cpu'))
got = [torch.zeros(1, dtype=torch.float64) for _ in range(2)]
torch.distributed.all_gather(got, torch.tensor([ms], dtype=torch.float64))
ok = ok and ms > 0 and float(got[0]) == float(got[1])
ok = ok and dist.choose_schedule(ms, 1e9) == '0' and dist.choose_schedule(ms, 1e-9) == 'tail'
open(os.path.join(os.environ['DP_OUT'], 'bucket' + str(rank) + '.txt'), 'w').write('OK' if ok else 'BAD')
'''


def test_gradient_buckets_gloo_world2(tmp_path):
    '''dist.GradBuckets (the opt-in overlapped all-reduce): pieces reduced as their hooks
    fire + the uncovered remainder in finish() == one all-reduce of the whole bucket'''
    script = tmp_path / 'bucket_worker.py'
    script.write_text(_BUCKET_WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', DP_OUT=str(tmp_path))
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    out = subprocess.run(
        [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
         '--master-addr', '127.0.0.1', '--master-port', str(port), str(script)],
        capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    for r in range(2):
        assert (tmp_path / ('bucket%d.txt' % r)).read_text() == 'OK'


def test_choose_schedule_threshold(monkeypatch):
    '''dist.choose_schedule: ONE threshold on measured all-reduce / measured step (default 5 %), NaN-safe'''
    from danet_amd import dist
    assert dist.TAIL_RATIO == 0.05
    assert dist.choose_schedule(0.09, 2.43) == '0'          # N = 8 over all links
    assert dist.choose_schedule(0.36, 2.43) == 'tail'       # N = 2
    assert dist.choose_schedule(0.46, 8.5) == 'tail' and dist.choose_schedule(0.40, 8.5) == '0'
    assert dist.choose_schedule(0.09, 2.43, ratio=0.03) == 'tail'
    assert dist.choose_schedule(float('nan'), 2.43) == '0' and dist.choose_schedule(0.5, 0.0) == '0'
    assert dist.measure_allreduce_ms(1000, 'cpu') == 0.0    # no process group: nothing to measure


def test_cli_flags_match_reference(hp):
    '''main.py:553-582 flag surface'''
    from danet_amd import cli
    a = cli.build_parser().parse_args(
        ['-n', 'x', '-m', 'valid', '-i', 'in', '-o', 'out', '-c', 'c.json', '-ne', '3', '-if', 'a.wav',
         '-ds', 'toy', '-lr', '0.01', '-tl', '64', '-bs', '8', '--no-save-on-epoch',
         '--no-valid-on-epoch'])
    assert (a.name, a.mode, a.input_pfile, a.output_pfile, a.config_file) == ('x', 'valid', 'in', 'out', 'c.json')
    assert (a.num_epoch, a.input_file, a.dataset, a.learn_rate, a.train_length, a.batch_size) == \
        (3, 'a.wav', 'toy', 0.01, 64, 8)
    assert a.no_save_on_epoch and a.no_valid_on_epoch
    assert 'kmeans' in hp.estimator_registry and hp.get_estimator('kmeans').USE_TRUTH is False
    assert hp.KMEANS_ITERS == 10


def test_library_never_reads_the_environment_and_options_abi(monkeypatch):
    '''ABI hygiene (include/danet_hip.h): variant selection is an option table set through
    danet_set_option, not the process environment; unknown names are errors; values read back;
    reset restores the defaults; the Python layer is what maps DANET_<NAME> onto the setter'''
    import ctypes
    from danet_amd import _lib
    lib = _lib.load()
    blob = open(_lib.LIB_PATH, 'rb').read()
    assert b'getenv' not in blob                # no import of getenv / secure_getenv at all
    names = _lib.option_names()
    assert len(names) == lib.danet_option_count() <= 16 and len(set(names)) == len(names)
    for must in ('gemm_dma', 'lstm_fwd_fused', 'lstm_bwd_u', 'lstm_spin_limit', 'gemm_yield'):
        assert must in names
    assert lib.danet_option_name(lib.danet_option_count()) is None
    lib.danet_reset_options()
    assert _lib.get_option('gemm_dma') == 3 and _lib.get_option('lstm_fwd_fused') == -1
    _lib.set_option('gemm_dma', 7)
    assert _lib.get_option('gemm_dma') == 7
    # the envelope query follows the option, not the environment
    monkeypatch.setenv('DANET_LSTM_FWD_FUSED', '0')
    assert lib.danet_lstm_fwd_fused_supported(128, 32, 300, 2, 600) == 1
    _lib.set_option('lstm_fwd_fused', 0)
    assert lib.danet_lstm_fwd_fused_supported(128, 32, 300, 2, 600) == 0
    monkeypatch.delenv('DANET_LSTM_FWD_FUSED')
    # ... and the Python layer applies the USER options' DANET_<NAME> through the setter; every other
    # option is an expert setting behind the ONE variable DANET_EXPERT="name=value,..."
    monkeypatch.setenv('DANET_GEMM_YIELD', '24')           # not a user switch any more: ignored
    monkeypatch.setenv('DANET_LSTM_SPIN_LIMIT', '4096')
    monkeypatch.setenv('DANET_EXPERT', 'gemm_yield=20, streamk=7,fork_spacer=0')
    _lib._expert = None
    _lib.apply_env_options()
    assert _lib.get_option('gemm_yield') == 20 and _lib.get_option('gemm_dma') == 3
    assert _lib.get_option('lstm_fwd_fused') == -1 and _lib.get_option('lstm_spin_limit') == 4096
    assert _lib.expert('streamk', 5) == 7 and _lib.expert('fork_spacer', True) is False
    assert _lib.expert('grouped_dw', 1) == 1 and _lib.expert('copy_stream', 'side') == 'side'
    for k in ('DANET_GEMM_YIELD', 'DANET_LSTM_SPIN_LIMIT', 'DANET_EXPERT'):
        monkeypatch.delenv(k)
    _lib._expert = None
    _lib.apply_env_options()
    assert _lib.get_option('gemm_yield') == 16 and _lib.get_option('lstm_spin_limit') == 0
    assert lib.danet_set_option(b'no_such_option', 1) == -1
    assert b'unknown option' in lib.danet_last_error()
    v = ctypes.c_int(0)
    assert lib.danet_get_option(b'no_such_option', ctypes.byref(v)) == -1


def test_options_are_thread_safe():
    '''two host threads flipping / reading options concurrently never see torn or foreign
    values (ints in atomics)'''
    import threading
    from danet_amd import _lib
    _lib.load()
    bad = []

    def worker(name, vals):
        for i in range(20000):
            _lib.set_option(name, vals[i % 2])
            if _lib.get_option(name) not in vals:
                bad.append((name, _lib.get_option(name)))
    ts = [threading.Thread(target=worker, args=('gemm_yield', (8, 16))),
          threading.Thread(target=worker, args=('gemm_wgs', (256, 512)))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not bad
    _lib.apply_env_options()
