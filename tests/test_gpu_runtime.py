'''
GPU tests (run with -m gpu): host threads and the hand-off status word.
Shared helpers: tests/gpu_helpers.py.
'''
import threading
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_two_host_threads_launch_concurrently():
    '''Two host threads, each on its own HIP stream with its own workspace, launch stream-K
    grouped GEMMs (the launches that number themselves from the process-wide atomic counter and
    hand partial tiles over through flags in their workspace) and persistent LSTM layers at the
    same time; every result equals the single-threaded one bit for bit.'''
    from danet_amd import ops, _lib
    _lib.load()
    dev = torch.device('cuda')
    rng = np.random.RandomState(0)
    M, N, K = 300, 1200, 4096
    A = torch.as_tensor(rng.randn(K, M).astype(np.float32)).cuda()
    Bm = torch.as_tensor(rng.randn(K, N).astype(np.float32)).cuda()
    T, B, D, H = 24, 16, 40, 64
    x = torch.as_tensor(rng.randn(T * B, D).astype(np.float32)).cuda()
    Ws = [torch.as_tensor((rng.randn(D + H, 4 * H) * 0.1).astype(np.float32)).cuda() for _ in range(2)]
    bs = [torch.as_tensor((rng.randn(4 * H) * 0.1).astype(np.float32)).cuda() for _ in range(2)]

    def work(n_iter):
        outs = []
        for _ in range(n_iter):
            C = torch.empty(M, N, device=dev)
            ops.gemm_group([(A, M, Bm, N, C, N, M, N, 0.0)], K, transA=True, max_workgroups=256)
            c = ops.lstm_layer_fwd(x, D, D, T, B, H, Ws, bs)
            outs.append((C, c.ypad[1:T + 1].clone()))
        return outs

    ref = work(1)[0]
    torch.cuda.synchronize()
    results, errors = {}, []

    def thread_main(tid):
        try:
            s = torch.cuda.Stream(device=dev)
            with torch.cuda.stream(s):
                results[tid] = work(12)
            s.synchronize()
        except Exception as e:      # noqa: BLE001
            errors.append(repr(e))

    ts = [threading.Thread(target=thread_main, args=(i,)) for i in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    torch.cuda.synchronize()
    assert not errors, errors
    assert ops.lstm_status_ok()
    for tid in (0, 1):
        for C, y in results[tid]:
            assert torch.equal(C, ref[0]) and torch.equal(y, ref[1])


def test_status_word_lives_in_pinned_host_memory_and_kernels_can_write_it(hp):
    '''single process: the persistent kernels' status word is pinned, device-mapped host
    memory (no per-step copy); a forced timeout is written by the GPU and read by the host
    directly, with the DANET_STATUS_TIMEOUT bit pattern (1.0f)'''
    from danet_amd import ops, _lib
    dev = torch.device('cuda')
    w = ops.status_word(dev)
    assert not w.is_cuda and w.is_pinned() and w.numel() == 4
    rng = np.random.RandomState(1)
    T, B, D, H = 8, 16, 16, 32
    x = torch.as_tensor(rng.randn(T * B, D).astype(np.float32)).cuda()
    Ws = [torch.as_tensor((rng.randn(D + H, 4 * H) * 0.1).astype(np.float32)).cuda() for _ in range(2)]
    bs = [torch.zeros(4 * H, device=dev) for _ in range(2)]
    ops.lstm_layer_fwd(x, D, D, T, B, H, Ws, bs)
    torch.cuda.synchronize()
    assert int(w[0]) == 0
    _lib.set_option('lstm_fault_inject', 1)
    _lib.set_option('lstm_spin_limit', 2048)
    ops.lstm_layer_fwd(x, D, D, T, B, H, Ws, bs)
    torch.cuda.synchronize()
    assert int(w[0]) == 0x3F800000
    assert w.view(torch.float32)[0].item() == 1.0
    assert not ops.lstm_status_ok() and int(w[0]) == 0
