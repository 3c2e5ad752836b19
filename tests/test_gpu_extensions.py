'''
GPU tests (-m gpu) of the SURVEY 8(f) rows built on top of the hot path:
f-1 k-means estimator (extension, no reference behaviour to match: validated by
convergence on separable embeddings), f-2 train loop / CLI pieces
(main.py:402-532, :551-750), f-4 parameter export under TF variable names.
'''
import json
import os

import numpy as np
import pytest
import torch

from oracle import danet_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _lstm_status():
    yield
    from danet_amd import ops
    torch.cuda.synchronize()
    assert ops.lstm_status_ok()


class _FakeModel(object):
    def __init__(self):
        self.vars = {}

    def get_variable(self, name, shape, init):
        if name not in self.vars:
            self.vars[name] = init(list(shape), torch.Generator().manual_seed(3)).cuda()
        return self.vars[name]


def test_kmeans_converges_to_cluster_means(hp):
    '''two well-separated embedding clusters: k-means attractors == cluster means
    == the truth-weighted attractors computed from the ideal assignment'''
    from danet_amd import modules, ops
    hp.load(dict(BATCH_SIZE=3, MAX_N_SIGNAL=2, EMBED_SIZE=20, NUM_ANCHOR=6, KMEANS_ITERS=8))
    hp.digest()
    rng = np.random.RandomState(0)
    B, T, F, E = 3, 40, 129, 20
    centres = rng.randn(B, 2, E) * 2.0
    label = rng.randint(0, 2, size=(B, T, F))
    embed = centres[np.arange(B)[:, None, None], label] + 0.05 * rng.randn(B, T, F, E)
    mix = rng.rand(B, T, F) + 0.5
    src_pwr = np.stack([(label == 0), (label == 1)], 1).astype(np.float64) + 0.1
    est = modules.KMeansEstimator(_FakeModel(), 'infer_estimator')
    assert est.USE_TRUTH is False
    cu = lambda a: torch.as_tensor(np.asarray(a)).to('cuda', torch.float32)
    attr = est(cu(embed), s_mix_pwr=cu(mix)).cpu().numpy()
    ref = O.est_truth_weighted(embed, src_pwr, mix)
    for b in range(B):                      # up to the attractor order
        d_same = np.abs(attr[b] - ref[b]).max()
        d_swap = np.abs(attr[b] - ref[b][::-1]).max()
        assert min(d_same, d_swap) < 1e-3 * np.abs(ref[b]).max(), (b, d_same, d_swap)


def test_kmeans_as_infer_estimator_in_model(hp):
    from danet_amd.model import Model
    hp.load(dict(BATCH_SIZE=2, MAX_N_SIGNAL=2, FFT_SIZE=64, FFT_STRIDE=16, EMBED_SIZE=8,
                 NUM_LSTM_LAYERS=1, LSTM_HDIM=16, NUM_ANCHOR=4, ENCODER_TYPE='bilstm-orig',
                 TRAIN_ESTIMATOR_METHOD='truth-weighted', INFER_ESTIMATOR_METHOD='kmeans',
                 SEPARATOR_TYPE='dot-softmax-orig'))
    hp.digest()
    model = Model('km', device='cuda').build()
    rng = np.random.RandomState(1)
    src = torch.as_tensor((rng.randn(2, 2, 12, 33) + 1j * rng.randn(2, 2, 12, 33)).astype(np.complex64)).cuda()
    v = model.valid_step(src)
    assert np.isfinite(float(v['loss'])) and np.isfinite(float(v['SNR']))
    sep = model.infer(src.sum(1))
    assert tuple(sep.shape) == (2, 2, 12, 33)
    assert np.abs((sep.sum(1) - src.sum(1)).cpu().numpy()).max() < 1e-4 * float(src.abs().max())


def test_cli_train_valid_test_debug_demo(hp, tmp_path, monkeypatch):
    '''reference command line end to end on the synthetic dataset'''
    import io
    import scipy.io.wavfile
    from danet_amd import cli, datasets
    monkeypatch.chdir(tmp_path)
    cfg = dict(BATCH_SIZE=2, MAX_N_SIGNAL=2, FFT_SIZE=64, FFT_STRIDE=16, EMBED_SIZE=8,
               NUM_LSTM_LAYERS=1, LSTM_HDIM=16, NUM_ANCHOR=4, MAX_TRAIN_LEN=24,
               ENCODER_TYPE='bilstm-orig', TRAIN_ESTIMATOR_METHOD='truth-weighted',
               INFER_ESTIMATOR_METHOD='anchor', SEPARATOR_TYPE='dot-softmax-orig',
               DATASET_TYPE='synth', LR_DECAY_TYPE='fixed', NUM_EPOCH_PER_LR_DECAY=1, DEBUG=True)
    (tmp_path / 'cfg.json').write_text(json.dumps(cfg))
    monkeypatch.setattr(datasets.SynthSpeechData, 'N_BATCH', {'train': 3, 'valid': 1, 'test': 1})
    monkeypatch.setattr(datasets.SynthSpeechData, 'N_FRAMES', 40)
    out = io.StringIO()
    model = cli.main(['-n', 'exp', '-m', 'train', '-c', 'cfg.json', '-ne', '2', '-lr', '0.001',
                      '-o', str(tmp_path / 'final')], out=out)
    log = out.getvalue()
    assert log.count(':') >= 6 and 'Epoch 2/2' in log and 'Valid  2/2' in log
    assert '[LR 0.001000 -> 0.000800]' in log                       # fixed decay every epoch
    assert os.path.exists('saves/exp_e1.npz') and os.path.exists('saves/exp_e2.npz')
    assert os.path.exists(str(tmp_path / 'final.npz'))
    saved = np.load('saves/exp_e2.npz')
    assert 'global/encoder/lstm0_fwd/LSTM/linear/W' in saved.files     # TF variable names
    assert 'global/encoder/output/W' in saved.files
    assert saved['global/encoder/lstm0_bwd/LSTM/linear/B'].shape == (64,)
    for k, v in model.param_dict().items():
        assert np.array_equal(saved[k], v), k
    # test mode from the checkpoint
    hp.reset()
    out = io.StringIO()
    cli.main(['-n', 'exp', '-m', 'test', '-c', 'cfg.json', '-i', 'saves/exp_e2'], out=out)
    assert 'Test: loss=' in out.getvalue()
    # debug mode
    hp.reset()
    out = io.StringIO()
    cli.main(['-n', 'exp', '-m', 'debug', '-c', 'cfg.json', '-i', 'saves/exp_e2'], out=out)
    dbg = np.load('debug/debug_data.npz')
    for k in ('embed', 'attrs', 'input', 'output', 'masks'):
        assert k in dbg.files
    # demo mode: wav in, C wavs out (16-bit 8 kHz -> resampled/STFT/iSTFT chain)
    hp.reset()
    wave = (datasets.speech_shaped_wave(np.random.RandomState(2), 4000, 8000)).astype(np.int16)
    scipy.io.wavfile.write('mix.wav', 8000, wave)
    out = io.StringIO()
    cli.main(['-n', 'exp', '-m', 'demo', '-c', 'cfg.json', '-i', 'saves/exp_e2', '-if', 'mix.wav'], out=out)
    for i in (1, 2):
        sr, y = scipy.io.wavfile.read('mix_separated_%d.wav' % i)
        assert sr == 8000 and len(y) == (1 + -(-4000 // 16)) * 16 and np.isfinite(y).all()
    # demo mode WITHOUT an input file (main.py:662-678): MAX_N_SIGNAL test utterances of
    # different lengths, zero-padded at random positions to a multiple of LENGTH_ALIGN,
    # summed, written to demo.wav, separated into demo_separated_{1,2}.wav
    hp.reset()
    (tmp_path / 'cfg2.json').write_text(json.dumps(dict(cfg, DATASET_TYPE='synth-varlen',
                                                         LENGTH_ALIGN=8)))
    monkeypatch.setattr(datasets.SynthVarLenSpeechData, 'MIN_FRAMES', 24)
    import random
    random.seed(5)
    out = io.StringIO()
    cli.main(['-n', 'exp', '-m', 'demo', '-c', 'cfg2.json', '-i', 'saves/exp_e2'], out=out)
    assert 'wrote demo.wav' in out.getvalue()
    sr, mix = scipy.io.wavfile.read('demo.wav')
    T = len(mix) // 16
    assert sr == 8000 and len(mix) == T * 16 and T % 8 == 0 and 24 <= T <= 40
    assert np.isfinite(mix).all() and np.abs(mix).max() > 0
    for i in (1, 2):
        sr, y = scipy.io.wavfile.read('demo_separated_%d.wav' % i)
        assert sr == 8000 and len(y) == len(mix) and np.isfinite(y).all()


def test_overlapped_allreduce_matches_single_allreduce_one_rank():
    '''opt-in DANET_OVERLAP_ALLREDUCE path under a real (1-rank) RCCL group:
    bit-identical parameters after 3 steps (own process: it owns a process group)'''
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import socket
    with socket.socket() as sk:
        sk.bind(('127.0.0.1', 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_PORT=str(port), MASTER_ADDR='127.0.0.1', RANK='0', WORLD_SIZE='1')
    # 50 steps: 'tail' + early optimizer step and the per-layer buckets stay BIT-EQUAL to the
    # default single all-reduce over a run long enough for Adam's state to matter
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_overlap_allreduce.py'), '50'],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0 and 'OK' in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]
    # ... and BASELINE configs[3] as written (4 x 600, C = 3, 142.5 MB bucket; the weight-gradient groups and
    # the hoisted forward take other kernel paths there): 8 steps, incl. the 'auto' decision
    out = subprocess.run([sys.executable, os.path.join(root, 'tools', 'check_overlap_allreduce.py'), '8', 'cfg4h600'],
                         capture_output=True, text=True, timeout=900, env=env)
    assert out.returncode == 0 and 'OK' in out.stdout, out.stdout[-1500:] + out.stderr[-1500:]
    assert "'schedule': '0'" in out.stdout
