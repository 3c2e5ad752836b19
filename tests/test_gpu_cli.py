'''
GPU tests (run with -m gpu): the drop-in train loop (cli / feed) on the GPU.
Shared helpers: tests/gpu_helpers.py.
'''
import random

import numpy as np
import pytest
from gpu_helpers import check_lstm_status, oracle_threads

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _lstm_status():
    with oracle_threads():
        yield
    check_lstm_status()


# ------------------------------------------------ data parallel, 2 ranks, the HIP path (one GPU)


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
