'''
Ops of the DANet hot path, MI355X-native: thin tensor-level wrappers over the
C ABI (include/danet_hip.h) plus the `torch.autograd.Function`s that let
PyTorch-ROCm drive backward.  Mirrors the roles of the reference's
`app/ops.py` (lyr_linear, lyr_lstm_flat, combinations, pit_mse_loss,
batch_snr) and of `Model.lyr_lstm` (main.py:76-132); names and argument meaning
follow the reference where a counterpart exists.

Everything here requires the HIP library and a GPU; nothing falls back to CPU.
'''
import itertools
import math

import torch

from . import _lib
from ._lib import ptr, check


def _L():
    return _lib.load()


def _f32(t):
    assert t.is_cuda and t.dtype == torch.float32, (t.device, t.dtype)
    return t


def _ws(nbytes, dev):
    w = _lib.workspace(nbytes, dev)
    return w, w.numel()


# ---------------------------------------------------------------------------
# raw wrappers (no autograd)
# ---------------------------------------------------------------------------
# stream-K schedule (danet_gemm_f32_streamk*: persistent workgroups, whole tiles data-parallel,
# the ragged last round cut along K, deterministic in-kernel fix-up) for single products on the
# critical path -- bit mask: 1 = dYc / unidirectional dX, 2 = output projection, 4 = the
# K-concatenated dX of a BiLSTM layer.  Default 5.  Round 3, in-step at cfg 2 (us per launch /
# ms per step): dX 133 -> 117 (3.273 -> 3.241), dYc 142 -> 125 (-> 3.265), both 3.228; the
# output projection (672 tiles, K = 600) LOSES on it (126 -> 132: two workgroups per CU instead
# of three, and 160 of its tiles need a fix-up) and stays on the tile kernel.
STREAMK = _lib.expert('streamk', 5)


def gemm(A, B, C, M, N, K, lda, ldb, ldc, transA=False, transB=False, bias=None, beta=0.0,
         max_workgroups=0, streamk=False, tag=None, stop_event=None):
    '''C[M,N] = op(A) op(B) (+bias) (+beta*C) on the fp32 matrix cores.
    A, B, C are tensors whose data_ptr() is element (0,0); ld* in elements.
    max_workgroups > 0 caps the launch (persistent workgroups); streamk selects the
    stream-K schedule (for a product that has the GPU to itself).'''
    L = _L()
    if streamk:
        need = _lib.ws_bytes(_lib.WS_GEMM_STREAMK, M, N, K)
        # dedicated (zero-initialised, never shared) scratch: it holds the stream-K
        # hand-off flags, which must only ever contain earlier launch sequence numbers
        w = _lib.workspace(need, C.device, tag='gemm_sk')
        with _lib.timed('gemm_f32', tag):
            if stop_event is not None:
                stop_event.arm()
            check(L.danet_gemm_f32_streamk(_lib.stream(), int(transA), int(transB), M, N, K,
                                           ptr(_f32(A)), lda, ptr(_f32(B)), ldb, ptr(_f32(C)), ldc,
                                           ptr(bias), float(beta), ptr(w), w.numel()))
            if stop_event is not None:
                stop_event.attached = True
        return C
    need = _lib.ws_bytes(_lib.WS_GEMM, M, N, K)
    w, wn = _ws(need, C.device)
    with _lib.timed('gemm_f32', tag):
        check(L.danet_gemm_f32(_lib.stream(), int(transA), int(transB), M, N, K,
                                  ptr(_f32(A)), lda, ptr(_f32(B)), ldb, ptr(_f32(C)), ldc,
                                  ptr(bias), float(beta), ptr(w), wn, int(max_workgroups)))
    return C


def gemm_kcat(A1, lda1, B1, ldb1, K1, A2, lda2, B2, ldb2, K2, C, M, N, ldc,
              transA=False, transB=False, bias=None, beta=0.0, tag=None, streamk=False,
              stop_event=None):
    '''C[M,N] = op(A1) op(B1) + op(A2) op(B2) (+bias) (+beta*C) in one launch.
    streamk: the hybrid stream-K schedule (no slabs / reduce kernel; K1 % 16 == 0)
    stop_event: a ForkEvent that completes with this launch (stream-K path only)'''
    L = _L()
    if streamk and K1 % 16 == 0:
        w = _lib.workspace(_lib.ws_bytes(_lib.WS_GEMM_STREAMK, M, N, K1 + K2), C.device,
                           tag='gemm_sk')
        with _lib.timed('gemm_f32', tag):
            if stop_event is not None:
                stop_event.arm()
            check(L.danet_gemm_f32_streamk_kcat(_lib.stream(), int(transA), int(transB), M, N,
                                                K1, ptr(_f32(A1)), lda1, ptr(_f32(B1)), ldb1,
                                                K2, ptr(_f32(A2)), lda2, ptr(_f32(B2)), ldb2,
                                                ptr(_f32(C)), ldc, ptr(bias), float(beta),
                                                ptr(w), w.numel()))
            if stop_event is not None:
                stop_event.attached = True
        return C
    # (no stream-K: two accumulating products on the tile kernel)
    gemm(A1, B1, C, M, N, K1, lda1, ldb1, ldc, transA=transA, transB=transB, bias=bias, beta=beta,
         tag=tag)
    gemm(A2, B2, C, M, N, K2, lda2, ldb2, ldc, transA=transA, transB=transB, beta=1.0, tag=tag)
    return C


# fp32 products on the bf16 matrix cores for products whose B operand is a weight (the output
# projection, dYc, dX; csrc/gemm_x6.hip): the weight is split into three bf16 pieces and laid out in
# the matrix instruction's operand order ONCE per optimizer step (`repack_weights`), the activation
# is split inside the kernel.  Same accuracy as the exact-fp32 kernels (six of the nine piece
# products), 1.6-1.8 x their speed at cfg 2.  DANET_GEMM_X6=0: the exact-fp32 kernels everywhere.
GEMM_X6 = int(__import__('os').environ.get('DANET_GEMM_X6', '3'))    # bit 0: weight products, bit 1: weight gradients


class _PackedWeight(object):
    __slots__ = ('W', 'N', 'K', 'sn', 'sk', 'lo', 'hi', 'out', 'stale', 'version', 'unused')

    def __init__(self, W, N, K, sn, sk):
        self.W, self.N, self.K, self.sn, self.sk = W, N, K, sn, sk
        self.lo = W.data_ptr()
        self.hi = self.lo + 4 * ((N - 1) * sn + (K - 1) * sk + 1)
        self.out = torch.empty(_lib.ws_bytes(_lib.WS_GEMM_PACK, N, K), dtype=torch.uint8,
                               device=W.device)
        self.stale, self.version, self.unused = True, -1, 0


_packs = {}      # device -> {(ptr, N, K, sn, sk): _PackedWeight}


def _pack_now(pws):
    arr = (_lib.GemmPack * len(pws))()
    for i, pw in enumerate(pws):
        arr[i].src, arr[i].stride_n, arr[i].stride_k = pw.W.data_ptr(), pw.sn, pw.sk
        arr[i].N, arr[i].K = pw.N, pw.K
        arr[i].out, arr[i].out_bytes = pw.out.data_ptr(), pw.out.numel()
    check(_L().danet_gemm_pack_weights(_lib.stream(), len(pws), arr))
    for pw in pws:
        pw.stale, pw.version = False, pw.W._version


def packed_weight(W, N, K, sn, sk):
    '''the operand-layout copy of the weight B(n, k) = W.flat[n * sn + k * sk] (W: a tensor whose
    data_ptr() is B(0, 0)), packed now if the weight has changed since its last pack'''
    reg = _packs.setdefault(_dev_key(W.device), {})
    key = (W.data_ptr(), N, K, sn, sk)
    pw = reg.get(key)
    if pw is None:
        pw = reg[key] = _PackedWeight(W, N, K, sn, sk)
    pw.unused = 0
    if pw.stale or pw.version != W._version:
        _pack_now([pw])
    return pw.out


def weights_written(t):
    '''a kernel of this library has written the parameter range `t` (torch's version counter
    does not see such writes): packs that overlap it are stale'''
    reg = _packs.get(_dev_key(t.device))
    if reg:
        lo = t.data_ptr()
        hi = lo + 4 * t.numel()
        for pw in reg.values():
            if pw.lo < hi and lo < pw.hi:
                pw.stale = True


def repack_weights(dev):
    '''re-pack every stale registered weight of the device in ONE launch on the current stream
    (the end of an optimizer step); packs nobody has asked for in 8 rounds are dropped'''
    reg = _packs.get(_dev_key(dev))
    if not reg:
        return
    todo = []
    for key in list(reg):
        pw = reg[key]
        pw.unused += 1
        if pw.unused > 8:
            del reg[key]
        elif pw.stale or pw.version != pw.W._version:
            todo.append(pw)
    if todo:
        _pack_now(todo)


def drop_packs(dev):
    '''forget every packed weight of the device (its parameters have moved)'''
    _packs.pop(_dev_key(dev), None)


def _x6_ok(M, N, *pairs):
    if not (GEMM_X6 & 1):
        return False
    for A, lda, K in pairs:
        if K % 4 or lda % 4 or lda < K or A.data_ptr() % 16 or M * lda >= (1 << 29):
            return False
    return True


def _x6_tn_ok(problems, K):
    if not (GEMM_X6 & 2) or len(problems) > 6:
        return False
    for pr in problems:
        A, lda, B, ldb = pr[:4]
        if len(pr) > 9 and pr[9] is not None:
            return False
        if lda % 4 or ldb % 4 or A.data_ptr() % 16 or B.data_ptr() % 16 or \
                K * max(lda, ldb) >= (1 << 29):
            return False
    return True


def gemm_w(A, lda, W, sn, sk, C, M, N, K, ldc, tag=None, stop_event=None,
           A2=None, lda2=0, W2=None, K2=0, sn2=None, sk2=None, bias=None):
    '''C[M,N] = A B^T (+ A2 B2^T) with B(n, k) = W.flat[n * sn + k * sk] a (packed) weight, on the
    bf16 matrix cores with fp32 accuracy (B2 with strides sn2 / sk2, default: B's).  The caller
    has checked `_x6_ok`.'''
    L = _L()
    p1 = packed_weight(W, N, K, sn, sk)
    p2 = packed_weight(W2, N, K2, sn if sn2 is None else sn2, sk if sk2 is None else sk2) if K2 else None
    need = _lib.ws_bytes(_lib.WS_GEMM_X6, M, N, K, K2)
    # dedicated (zero-initialised, never shared) scratch: its first 16 KB are the K-slice tickets
    w = _lib.workspace(need, C.device, tag='gemm_x6')
    wn = w.numel()
    with _lib.timed('gemm_x6', tag):
        if stop_event is not None:
            stop_event.arm()
        check(L.danet_gemm_x6(_lib.stream(), M, N, K, ptr(_f32(A)), lda, ptr(p1),
                              K2, ptr(_f32(A2)) if K2 else None, lda2, ptr(p2) if K2 else None,
                              ptr(_f32(C)), ldc, ptr(bias), ptr(w), wn))
        if stop_event is not None:
            stop_event.attached = True
    return C


def _x6_tn_tiles(M, N):
    '''128 x 128 tiles of one product of danet_gemm_x6_tn_grouped (include/danet_hip.h): 1..4 rows
    beyond a multiple of 128 are tail rows, not a tile row (N a multiple of 4)'''
    tail = M % 128
    rows = M - tail if (M > 128 and 1 <= tail <= 4 and N % 4 == 0) else M
    return ((rows + 127) // 128) * ((N + 127) // 128)


def gemm_group(problems, K, transA=False, transB=False, max_workgroups=0):
    '''up to 6 products sharing K and the transpose flags as ONE stream-K launch.
    problems: list of (A, lda, B, ldb, C, ldc, M, N, beta) with tensors whose data_ptr()
    is element (0,0); an optional 10th entry is the bias vector.'''
    L = _L()
    x6 = transA and not transB and _x6_tn_ok(problems, K)
    arr = (_lib.GemmProblem * len(problems))()
    keep = []
    for i, pr in enumerate(problems):
        A, lda, B, ldb, C, ldc, M, N, beta = pr[:9]
        bias = pr[9] if len(pr) > 9 else None
        A, B, C = _f32(A), _f32(B), _f32(C)
        keep += [A, B, C, bias]
        arr[i].A, arr[i].lda, arr[i].B, arr[i].ldb = ptr(A), lda, ptr(B), ldb
        arr[i].C, arr[i].ldc, arr[i].M, arr[i].N = ptr(C), ldc, M, N
        arr[i].bias, arr[i].beta = ptr(bias), float(beta)
    dev = problems[0][4].device
    if x6:
        # both operands split inside the kernel (csrc/gemm_x6.hip, TN section); not persistent:
        # `max_workgroups` does not apply
        need = _lib.ws_bytes(_lib.WS_GEMM_X6_TN, sum(pr[6] * pr[7] for pr in problems),
                             sum(_x6_tn_tiles(pr[6], pr[7]) for pr in problems), K)
        w = _lib.workspace(need, dev, tag='gemm_x6_tn') if need else None
        with _lib.timed('gemm_x6_tn_group'):
            check(L.danet_gemm_x6_tn_grouped(_lib.stream(), K, len(problems), arr, ptr(w),
                                             w.numel() if w is not None else 0))
        return
    w = _lib.workspace(_lib.ws_bytes(_lib.WS_GEMM_STREAMK, 0, 0, K), dev, tag='gemm_sk')
    with _lib.timed('gemm_f32_group'):
        check(L.danet_gemm_f32_streamk_grouped(_lib.stream(), int(transA), int(transB), K,
                                               len(problems), arr, int(max_workgroups),
                                               ptr(w), w.numel()))


# weight gradients of a BiLSTM layer as one grouped stream-K launch (DANET_EXPERT grouped_dw=0:
# four split-K launches + reduce kernels)
GROUPED_DW = _lib.expert('grouped_dw', 1)
GROUPED_DW_WGS = _lib.expert('grouped_dw_wgs', 256)
GROUPED_GX = _lib.expert('grouped_gx', 512)   # grid of the grouped gx launch (0: two launches on two streams)


def colsum(A, M, N, lda, out, beta=0.0):
    L = _L()
    w, wn = _ws(_lib.ws_bytes(_lib.WS_COLSUM, M, N), out.device)
    check(L.danet_colsum_f32(_lib.stream(), M, N, ptr(_f32(A)), lda, ptr(out), float(beta),
                             ptr(w), wn))
    return out


def center(x, B, T, D, in_layout, ld_in, out, out_layout, ld_out):
    '''out[b] = x[b] - mean(x[b]) with an optional layout switch; returns means [B]'''
    L = _L()
    scratch = torch.empty(_lib.ws_bytes(_lib.WS_CENTER_MEAN, B) // 4, dtype=torch.float32, device=x.device)
    check(L.danet_center(_lib.stream(), B, T, D, ptr(_f32(x)), in_layout, ld_in,
                         ptr(out), out_layout, ld_out, ptr(scratch)))
    return scratch[:B]


def frontend(src, want_phase=False, want_mix=False):
    '''main.py:233-240.  src complex64 [B,C,T,F] ->
    dict(src_pwr [B,C,T,F], mix_pwr, mix_log [B,T,F], phasor [B,T,F,2], ...)'''
    assert src.is_cuda and src.dtype == torch.complex64
    src = src.contiguous()
    B, C, T, F = src.shape
    N = T * F
    dev = src.device
    out = dict(
        src_pwr=torch.empty(B, C, T, F, device=dev),
        mix_pwr=torch.empty(B, T, F, device=dev),
        mix_log=torch.empty(B, T, F, device=dev),
        phasor=torch.empty(B, T, F, 2, device=dev))
    phase = torch.empty(B, T, F, device=dev) if want_phase else None
    mix = torch.empty(B, T, F, dtype=torch.complex64, device=dev) if want_mix else None
    check(_L().danet_frontend_fwd(
        _lib.stream(), B, C, N, ptr(torch.view_as_real(src)), ptr(out['mix_pwr']),
        ptr(out['mix_log']), ptr(out['phasor']), ptr(phase), ptr(out['src_pwr']),
        ptr(torch.view_as_real(mix)) if mix is not None else None))
    if want_phase:
        out['phase'] = phase
    if want_mix:
        out['mix'] = mix
    return out


def reattach_phase(sep_pwr, phasor, perm_idx=None):
    '''main.py:281-284 / :330-335 (with the permutation gather of :293-306)'''
    B, C, T, F = sep_pwr.shape
    out = torch.empty(B, C, T, F, dtype=torch.complex64, device=sep_pwr.device)
    check(_L().danet_reattach_phase(
        _lib.stream(), B, C, T * F, ptr(_f32(sep_pwr.contiguous())), ptr(phasor),
        ptr(perm_idx), ptr(torch.view_as_real(out))))
    return out


def combinations(s_data, subset_size, total_size=None, name=None):
    '''reference app/ops.py:273-292: gather rows by itertools.combinations'''
    if total_size is None:
        total_size = s_data.shape[0]
    combs = torch.tensor(list(itertools.combinations(range(total_size), subset_size)),
                         device=s_data.device)
    return s_data[combs]


def stft(x, window, fft_size, fft_stride):
    '''x float32 [n_sig, Ls] (or [Ls]) -> complex64 [n_sig, T, F]
    (app/utils.py:117-122).  Raises ValueError when Ls < fft_size like scipy.'''
    squeeze = x.dim() == 1
    if squeeze:
        x = x[None]
    x = _f32(x.contiguous())
    n_sig, Ls = x.shape
    L = _L()
    T = L.danet_stft_num_frames(Ls, fft_size, fft_stride)
    if T < 0:
        raise ValueError('window is longer than input signal')
    F = fft_size // 2 + 1
    out = torch.empty(n_sig, T, F, dtype=torch.complex64, device=x.device)
    check(L.danet_stft(_lib.stream(), n_sig, Ls, fft_size, fft_stride, ptr(x),
                       ptr(_f32(window)), ptr(torch.view_as_real(out))))
    return out[0] if squeeze else out


def istft(X, stride, window):
    '''utils.istft (app/utils.py:53-75).  X complex64 [n_sig, T, F] or [T, F]
    -> float64 [n_sig, T*stride]'''
    squeeze = X.dim() == 2
    if squeeze:
        X = X[None]
    assert X.dtype == torch.complex64 and X.is_cuda
    X = X.contiguous()
    n_sig, T, F = X.shape
    N = (F - 1) * 2
    L = _L()
    out = torch.empty(n_sig, T * stride, dtype=torch.float64, device=X.device)
    w, wn = _ws(_lib.ws_bytes(_lib.WS_ISTFT, n_sig, T, N, stride), X.device)
    check(L.danet_istft(_lib.stream(), n_sig, T, N, stride, ptr(torch.view_as_real(X)),
                        ptr(_f32(window)), ptr(out), ptr(w), wn))
    return out[0] if squeeze else out


# ---------------------------------------------------------------------------
# LSTM layer (both directions), raw forward / backward on time-major tensors
# ---------------------------------------------------------------------------
class _LayerCtx(object):
    __slots__ = ('x', 'ldx', 'D', 'T', 'B', 'H', 'ndir', 'ypad', 'gates', 'cells',
                 'Ws', 'bs')


# ---- hand-off status of the persistent kernels + bounded host run-ahead -------
# ONE sticky 4-byte word per device (include/danet_hip.h: `status` of danet_lstm_fwd/bwd):
# the kernels store DANET_STATUS_TIMEOUT (the bit pattern of 1.0f) into it when a bounded
# inter-workgroup wait times out.  Where it lives:
#   * single process: in PINNED, device-mapped host memory -- the kernels write it across
#     the bus (only on a timeout; they read it once per 256 failed polls), the host reads
#     it directly once a step's event has completed.  No copy, no extra launch.
#   * data parallel: Model points it at the 4 spare floats behind its flat gradient bucket,
#     so the word rides in the step's gradient all-reduce (SUM of 1.0f per timed-out rank):
#     after the reduction every rank holds the same value and they all raise at the same
#     step instead of one rank leaving the others hanging in the next collective.  The
#     reduced word is copied to pinned memory asynchronously once per step.
# `StepFence` bounds how far the host may run ahead of the GPU: Model.train_step /
# valid_step / infer record an event when a step has been enqueued and wait for the event
# of the step MAX_STEPS_IN_FLIGHT back before enqueuing the next.  A step whose event has
# completed is "retired": its status is inspected then, so a timeout raises DanetHipError
# at most MAX_STEPS_IN_FLIGHT steps later (round 2 polled every 16 calls).  Bounding the
# run-ahead also keeps the host from ever sitting 8-9 steps (30 ms of queued kernels)
# ahead of the GPU -- the situation in which the one-off 13-19 ms recurrent-kernel stall
# of DESIGN.md 5 was observed.
import collections as _collections
import os as _os

MAX_STEPS_IN_FLIGHT = int(_os.environ.get('DANET_MAX_STEPS_IN_FLIGHT', '4'))
STATUS_HOST = _os.environ.get('DANET_STATUS_HOST', '1') == '1'
TIMEOUT_MSG = ('persistent LSTM kernel: an inter-workgroup hand-off timed out -- the outputs of '
               'that launch (and everything computed from them) are invalid')


class _DeviceStatus(object):
    __slots__ = ('dev', 'word', 'own', 'host_mapped', 'slots', 'queue', 'n', 'retired', 'events')

    def __init__(self, dev):
        self.dev = dev
        if STATUS_HOST:
            self.own = torch.zeros(4, dtype=torch.int32).pin_memory()
        else:
            self.own = torch.zeros(4, dtype=torch.int32, device=dev)
        self.word, self.host_mapped = self.own, STATUS_HOST
        self.slots = torch.zeros(MAX_STEPS_IN_FLIGHT + 2, 4, dtype=torch.int32).pin_memory()
        self.queue = _collections.deque()     # (event, slot or -1, may_raise)
        self.n = 0
        self.retired = 0
        self.events = []                      # ring of re-recorded step events (created once)


_status = {}


def _dev_status(dev):
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _status.get(key)
    if st is None:
        st = _status[key] = _DeviceStatus(torch.device('cuda', key))
    return st


def status_word(dev):
    '''the device's sticky 4 x int32 status tensor (element 0 = LSTM hand-off) that the
    launches are given; pinned host memory or device memory (see above)'''
    return _dev_status(dev).word


def set_status_word(dev, word=None):
    '''point the device's status word at caller-owned DEVICE memory (an int32 view of 4
    elements; Model uses the tail of its gradient bucket under data parallelism); None
    restores the process-owned word'''
    st = _dev_status(dev)
    torch.cuda.synchronize(st.dev)
    st.queue.clear()
    if word is None:
        st.word, st.host_mapped = st.own, STATUS_HOST
    else:
        assert word.is_cuda and word.dtype == torch.int32 and word.numel() == 4
        st.word, st.host_mapped = word, False


def _word_flag(st):
    '''current value of word 0 (caller has synchronised what it needs)'''
    return int(st.word[0].item())


def _raise_timeout(st):
    torch.cuda.synchronize(st.dev)
    st.word.zero_()              # so the caller may restore parameters and go on
    st.queue.clear()
    raise _lib.DanetHipError(TIMEOUT_MSG)


def _retire(st, block):
    ev, slot, may_raise = st.queue[0]
    if block:
        ev.synchronize()
    elif not ev.query():
        return False
    st.queue.popleft()
    st.retired += 1
    flag = int(st.slots[slot, 0]) if slot >= 0 else int(st.word[0])
    if flag != 0 and may_raise:
        _raise_timeout(st)
    return True


def _collective():
    return torch.distributed.is_available() and torch.distributed.is_initialized()


def poll_status(dev):
    '''admission of a new step (called first by Model.train_step / valid_step / infer):
    retire every completed step, wait until fewer than MAX_STEPS_IN_FLIGHT are in flight;
    raises DanetHipError when a retired step recorded a hand-off timeout.

    Under torch.distributed the retirement is DETERMINISTIC: at the admission of step n a rank
    retires exactly step n - MAX_STEPS_IN_FLIGHT (blocking on its event) and nothing else -- no
    opportunistic `query()` path, whose outcome depends on how far each rank's GPU happens to
    be.  Every rank issues the same sequence of steps and the flag of step k is the same on
    every rank (it rides in that step's gradient all-reduce), so all ranks raise at the
    admission of the SAME step, with the same collectives enqueued behind it
    (test_handoff_timeout_raises_at_the_same_step_on_every_rank_gloo_world2).'''
    st = _dev_status(dev)
    if not _collective():
        while st.queue and _retire(st, block=False):
            pass
    while len(st.queue) >= max(MAX_STEPS_IN_FLIGHT, 1):
        _retire(st, block=True)


def _record_event(st):
    '''the event of the step just enqueued, from a ring of MAX_STEPS_IN_FLIGHT + 2 events that
    are created once and re-recorded: an entry is reused only after the step it stood for has
    been retired (the queue never holds more than MAX_STEPS_IN_FLIGHT entries)'''
    ring = getattr(st, 'events', None)
    if ring is None:                          # (tests build the record by hand)
        ring = st.events = []
    n = max(MAX_STEPS_IN_FLIGHT, 1) + 2
    if len(ring) < n:
        ring.append(torch.cuda.Event())
        ev = ring[-1]
    else:
        ev = ring[st.n % n]
    ev.record()
    return ev


def step_done(dev, collective_consistent=True):
    '''a step has been enqueued on the current stream: record its event (and, for a
    device-resident word, an asynchronous 16-byte snapshot into pinned memory).
    collective_consistent=False (data parallel, a step without a gradient all-reduce): the
    word is rank-local, so retiring the step does not raise -- `check_status` does, on
    every rank together.'''
    st = _dev_status(dev)
    slot = -1
    if not st.host_mapped:
        slot = st.n % st.slots.shape[0]
        st.slots[slot].copy_(st.word, non_blocking=True)
    st.n += 1
    st.queue.append((_record_event(st), slot, collective_consistent))


def check_status(dev=None):
    '''blocking check (end of an epoch / a sweep, before results are written, tests);
    raises DanetHipError on a recorded timeout.  Under torch.distributed the flag is
    MAX-reduced first, so every rank raises (or none does).'''
    for st in ([_dev_status(dev)] if dev is not None else list(_status.values())):
        torch.cuda.synchronize(st.dev)
        st.queue.clear()
        flag = 1 if _word_flag(st) != 0 else 0
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            t = torch.tensor([flag], dtype=torch.int32, device=st.dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            flag = int(t.item())
        if flag != 0:
            _raise_timeout(st)


def lstm_status_ok():
    '''True when no persistent-LSTM launch since the last call reported an
    inter-workgroup timeout (clears the word).  Synchronises.'''
    ok = True
    for st in _status.values():
        torch.cuda.synchronize(st.dev)
        st.queue.clear()
        if _word_flag(st) != 0:
            ok = False
            st.word.zero_()
    return ok


def _lstm_ws(T, B, H, ndir, dev):
    n = _lib.ws_bytes(_lib.WS_LSTM, T, B, H, ndir)
    return torch.empty(n, dtype=torch.uint8, device=dev), n


# ---- concurrent chains on side HIP streams ---------------------------------
# The GEMMs of the two scan directions (and dWx / dWh / dX in backward) are
# independent and individually too small to fill 256 CUs evenly (e.g. 320 tiles);
# issued on separate streams their tiles co-schedule.  All tensors are allocated
# on the main stream and the main stream joins the side streams before anything
# is returned, so the caching allocator never recycles memory still in use.
_side = {}
SIDE_STREAMS = int(__import__('os').environ.get('DANET_SIDE_STREAMS', '1'))
# persistent-workgroup cap for GEMMs that run under a BPTT kernel (per chain; 0 = off).
# Measured at cfg 2: caps of 32..96 all LOSE (5.4-7.1 ms/step vs 5.2 uncapped): the
# interference is fabric contention on the exchange hops, not CU placement.
OVERLAP_GEMM_WGS = _lib.expert('overlap_gemm_wgs', 0)


def _side_streams(dev, n):
    # (stream priorities were measured in round 6 -- side streams at priority 1 / -1 against the main
    # stream's 0: 2.402 / 2.399 against 2.401 ms per cfg-2 step, BPTT 316.0 / 315.6 against 315.6 us: the
    # weight-gradient group's cost to the BPTT kernel is not a matter of dispatch priority)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    pool = _side.setdefault(key, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:n]


_copy = {}
# which stream uploads ride on (feed.BatchFeed): 'side' = the first side stream, 'own' = a stream
# of their own.  HIP maps streams onto a handful of hardware queues; a third busy stream lands on
# the main or the side stream's queue depending on creation order and the kernels of the two then
# serialise: the same loop measured 3.14 or 4.2 ms per cfg-2 step from one stream object to the
# next (tools/feed_probe.py, profiles/r04_feed_probe.txt).  The side stream exists anyway and is
# idle at the step boundary, where the feed issues the upload of the next batch.
COPY_STREAM = _lib.expert('copy_stream', 'side')


def copy_stream(dev):
    '''the upload stream of a device (created once per process)'''
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    st = _copy.get(key)
    if st is None:
        sides = _side_streams(dev, max(SIDE_STREAMS, 1))
        st = _copy[key] = sides[0] if (COPY_STREAM == 'side' and SIDE_STREAMS > 0) else \
            torch.cuda.Stream(device=dev)
    return st


def prepare_streams(dev):
    '''Create the process's HIP context and the side streams NOW.  Must run before an
    RCCL communicator is created: a group initialised first takes the hardware queues the
    side streams would get, the chains then share a queue with the main stream and the
    train step is 0.6 ms (16 %) slower (tools/dist_order_probe.py: 4.32 vs 3.72 ms).'''
    torch.zeros(1, device=dev)
    _side_streams(dev, max(SIDE_STREAMS, 1))
    copy_stream(dev)
    torch.cuda.synchronize(dev)


# An event that rides on a stream-K launch's own dispatch packet (danet_next_launch_events)
# instead of a hipEventRecord behind it: the record costs the launching stream ~4.4 us before its
# next kernel, the attached event ~1.1 us, and the side chain starts ~3.7 us earlier
# (tools/csrc/event_gap.hip).  Raw events from a small rotating pool per device (a slot is reused
# dozens of launches later; the host keeps at most MAX_STEPS_IN_FLIGHT steps queued).
FORK_ATTACH = _lib.expert('fork_attach', True)
_fork_event_pool = {}


class ForkEvent(object):
    __slots__ = ('handle', 'attached')

    def __init__(self, handle):
        self.handle, self.attached = handle, False

    def arm(self):
        '''the NEXT stream-K launch of this host thread completes the event; `attached` is set
        by the caller once that launch has been accepted (a rejected launch consumes the armed
        event inside the library and nothing may wait for it)'''
        check(_L().danet_next_launch_events(None, self.handle))


def fork_event(dev):
    '''the next ForkEvent of the device's pool, or None when attaching is switched off'''
    if not FORK_ATTACH or SIDE_STREAMS <= 0:
        return None
    pool = _fork_event_pool.setdefault(str(dev), {'i': 0, 'ev': []})
    if len(pool['ev']) < 32:
        h = _lib.c_p()
        check(_L().danet_event_create(__import__('ctypes').byref(h)))
        pool['ev'].append(h.value)
        return ForkEvent(h.value)
    pool['i'] = (pool['i'] + 1) % 32
    return ForkEvent(pool['ev'][pool['i']])


# A deferred side chain (a weight-gradient group) and the main stream's next kernel (the next layer's
# persistent BPTT kernel) become runnable at the same instant: when the launch the fork event rides on
# completes.  Nothing orders the two dispatches, and in ~4 % of the launches the group got its 480
# workgroups onto the CUs first: the BPTT kernel, whose workgroups exchange partials every time step,
# then starts piecemeal and takes 470-500 us instead of 360 (tools/bptt_hist.sh).  One tiny kernel
# in front of the group on the SIDE stream (a few microseconds of dispatch + run) lets the BPTT
# kernel's workgroups go first; it costs the main stream nothing.  DANET_EXPERT fork_spacer=0 switches it off.
FORK_SPACER = _lib.expert('fork_spacer', True)
_spacer_buf = {}


def _spacer(dev):
    '''one tiny kernel of THIS library on the current (side) stream -- danet_leaky_relu over 64
    floats, in place; until round 5 a `tensor.zero_()`, the only torch kernel left in a train step'''
    t = _spacer_buf.get(_dev_key(dev))
    if t is None:
        t = _spacer_buf[_dev_key(dev)] = torch.zeros(64, device=dev)
    check(_L().danet_leaky_relu(_lib.stream(), 64, ptr(t), None, 0.0, ptr(t)))


class _Fork(object):
    '''with _Fork(dev, n) as f:  f.run(i, fn)  -> fn runs on chain i
    (chain 0 = the current stream, chain i>0 = side stream i-1); join on exit.'''

    def __init__(self, dev, nchains, defer=False, keep=(), lazy=False, event=None):
        '''defer=True: do not join on exit -- the side chains keep running under
        whatever the main stream does next (e.g. weight-gradient GEMMs under the
        next layer's latency-bound BPTT kernel, which leaves most CUs idle);
        `join_deferred()` joins them.  `keep` = tensors the chains read that the
        caller is about to drop (kept alive until the join).
        lazy=True: the fork event is recorded when the first side chain is issued instead of
        on entry -- a fork that ends up with no side work then costs the main stream nothing
        (an event record is a ~7 us bubble in front of the next kernel); only for forks whose
        main-stream work comes AFTER their side chains.
        event: a ForkEvent already attached to the last launch the side chains have to wait for
        (then no event is recorded at all).'''
        self.main = torch.cuda.current_stream(dev)
        n = min(nchains - 1, SIDE_STREAMS)
        self.sides = _side_streams(dev, n) if n > 0 else []
        self.used = set()
        self.defer, self.keep = defer, keep
        self.lazy, self.forked = lazy, False
        self.event = event if (event is not None and event.attached) else None

    def _fork_now(self):
        if self.sides and not self.forked:
            if self.event is not None:
                for s in self.sides:
                    check(_L().danet_stream_wait_event(s.cuda_stream, self.event.handle))
            else:
                ev = self.main.record_event()
                for s in self.sides:
                    s.wait_event(ev)
        self.forked = True

    def __enter__(self):
        if not self.lazy:
            self._fork_now()
        return self

    def run(self, chain, fn):
        if not self.sides or chain == 0:
            return fn()
        self._fork_now()
        s = self.sides[(chain - 1) % len(self.sides)]
        self.used.add(s)
        with torch.cuda.stream(s):
            q = _lazy.get(_dev_key(s.device))
            if q:
                self.keep = tuple(self.keep) + tuple(k for _f, k in q)
                _flush_lazy(s.device)  # queued small kernels ride on this fork's event
            elif self.defer and self.event is not None and FORK_SPACER:
                _spacer(s.device)
            return fn()

    def after_all(self, fn, wait_main=False):
        '''run fn() on one of the used side streams once ALL side chains of this
        fork have been issued (that stream first waits for the others) -- the
        place to launch work that consumes everything the chains produce, e.g. a
        gradient-bucket all-reduce.  wait_main: part of that work was issued on the
        main stream (chain 0), so the side stream waits for the main stream's
        current position too.  With no side stream in use, fn runs in place.'''
        used = list(self.used)
        if not used:
            return fn()
        first = used[0]
        for s in used[1:]:
            first.wait_stream(s)
        if wait_main:
            first.wait_stream(self.main)
        with torch.cuda.stream(first):
            return fn()

    def __exit__(self, *exc):
        if self.defer and self.used:
            _deferred.append((self.main, tuple(self.used), self.keep))
        else:
            for s in self.used:
                self.main.wait_stream(s)
        return False


_deferred = []

# Called as hook(tag, params) when the gradients of `params` have been fully
# issued (on the stream that is current during the call); tag = ('layer', l) /
# ('out',).  Model uses it to launch per-bucket gradient all-reduces that overlap
# the rest of backward (dist.GradBuckets).
# ('rest', ) additionally fires on a side stream that has waited for the main stream right
# after the BOTTOM layer's BPTT kernel of an encoder was issued: from then on every gradient
# except the bottom layer's own is final (dist.TailOverlap reduces them under the bottom
# layer's weight-gradient GEMMs).  Entries are weak references to bound methods (a
# disposed Model drops out by itself).
GRAD_READY_HOOKS = []


def add_grad_ready_hook(bound_method):
    import weakref
    GRAD_READY_HOOKS.append(weakref.WeakMethod(bound_method))


def _live_hooks():
    live = [(r, r()) for r in GRAD_READY_HOOKS]
    if any(h is None for _, h in live):
        GRAD_READY_HOOKS[:] = [r for r, h in live if h is not None]
    return [h for _, h in live if h is not None]


def _fire_grad_ready(tag, params):
    for h in _live_hooks():
        h(tag, params)


# Fast backward (scoped: only inside `with ops.fast_backward():`, which Model.train_step
# enters).  (1) kernels ADD the parameter gradients straight into an existing dense
# `param.grad` (a view of Model's flat all-reduce bucket) and hand autograd None -- one
# elementwise accumulate kernel less per parameter; outside the scope every backward returns
# ordinary gradient tensors, so torch.autograd.grad / double backward / foreign optimisers see
# standard autograd behaviour.  (2) the gradient-ready hooks above fire.
DIRECT_GRADS = _lib.expert('direct_grads', True)
_fast_depth = [0]


class fast_backward(object):
    def __enter__(self):
        _fast_depth[0] += 1
        return self

    def __exit__(self, *exc):
        _fast_depth[0] -= 1
        return False


def _fast():
    return _fast_depth[0] > 0


def _grad_target(param, shape, dev):
    """(tensor to accumulate the gradient of `param` into, is_direct).  Direct =
    inside `fast_backward()` and the parameter already owns a dense .grad (e.g. a
    view of Model's flat gradient bucket): kernels add into it (beta=1) and
    autograd is handed None."""
    g = param.grad if (DIRECT_GRADS and _fast() and torch.is_tensor(param)) else None
    if g is not None and g.is_contiguous() and tuple(g.shape) == tuple(shape):
        return g, True
    return torch.empty(*shape, device=dev), False


def join_deferred():
    '''main stream waits for every deferred side chain -- ONE wait per distinct side stream
    (each wait is an event record + a barrier packet: four of them in a row put a 17 us bubble
    in front of the optimiser kernel)'''
    _flush_lazy()                      # nothing forked since they were queued: run them here
    waits = {}
    while _deferred:
        main, used, _keep = _deferred.pop()
        for s in used:
            waits.setdefault((main.cuda_stream, s.cuda_stream), (main, s))
    for main, s in waits.values():
        main.wait_stream(s)


def lstm_prefill_fwd(T, B, ldy, ypads, wss):
    '''ONE fill launch for the output buffers (+ workspaces) of several forward launches'''
    n = len(ypads)
    yp = (_lib.c_p * n)(*[ptr(y) for y in ypads])
    wp = (_lib.c_p * n)(*[ptr(w) for w in wss])
    check(_L().danet_lstm_fwd_prefill(_lib.stream(), T, B, ldy, n, yp, wp))


# the input's centring and the recurrent launches' prefill in one launch (danet_encoder_prologue)
ENC_PROLOGUE = _lib.expert('enc_prologue', True)


def encoder_prologue(x, B, T, F, xc, Fp, H, ndir, ypads, fwd_wss, bwd_wss):
    '''xc[T, B, Fp] = x[B, T, F] - mean_{t,f}(x) (time-major, zero pad) AND the prefill of the n recurrent
    launches' buffers (bwd_wss: + the BPTT rings), one launch; False: the rings were NOT prefilled'''
    n = len(ypads)
    yp = (_lib.c_p * n)(*[ptr(y) for y in ypads])
    fp = (_lib.c_p * n)(*[ptr(w) for w in fwd_wss])
    bp = (_lib.c_p * n)(*[ptr(w) for w in bwd_wss]) if bwd_wss is not None else None
    scratch = torch.empty(_lib.ws_bytes(_lib.WS_CENTER_MEAN, B) // 4, dtype=torch.float32, device=x.device)
    args = (_lib.stream(), B, T, F, ptr(_f32(x)), 0, F, ptr(xc), 1, Fp, ptr(scratch), H, ndir, ndir * H, n, yp, fp)
    rc = _L().danet_encoder_prologue(*args, bp)
    if rc == -3 and bp is not None:      # BPTT geometry outside the reduce-scatter kernel: it prefills itself
        check(_L().danet_encoder_prologue(*args, None))
        return False
    check(rc)
    return True


def lstm_prefill_train(T, B, H, ndir, ypads, fwd_wss, bwd_wss):
    '''ONE fill launch for the forward launches' buffers AND the partial-dh rings of the BPTT launches
    that will follow; False (nothing launched) when the shape takes the all-gather BPTT kernel, which
    prefills its own exchange medium'''
    n = len(ypads)
    yp = (_lib.c_p * n)(*[ptr(y) for y in ypads])
    fp = (_lib.c_p * n)(*[ptr(w) for w in fwd_wss])
    bp = (_lib.c_p * n)(*[ptr(w) for w in bwd_wss])
    return _L().danet_lstm_train_prefill(_lib.stream(), T, B, H, ndir, ndir * H, n, yp, fp, bp) == 0


def lstm_layer_fwd(x, ldx, D, T, B, H, Ws, bs, x_pad_zero=False, ypad=None, ws=None):
    '''x: time-major [T*B rows, ldx] tensor (data_ptr = row 0), D valid columns.
    Ws[d]: [D+H, 4H] (reference layout, rows 0..D-1 input, D.. recurrent),
    bs[d]: [4H].  Returns ctx with ypad [T+2, B, ndir*H].
    x_pad_zero: the caller guarantees that columns D..ldx-1 of x hold finite values (zeros).
    The fused-input kernel loads whole float4 groups of a row and masks only W for k >= D,
    so with D % 4 != 0 a non-finite pad value would poison the gates (0 * NaN); without the
    guarantee such inputs take the hoisted GEMM path, which never reads the pad.'''
    ndir = len(Ws)
    dev = x.device
    L = _L()
    gates = [torch.empty(T * B, 4 * H, device=dev) for _ in range(ndir)]
    cells = [torch.empty(T * B, H, device=dev) for _ in range(ndir)]
    flags = 0
    if ypad is None:
        ypad = torch.empty(T + 2, B, ndir * H, device=dev)
        ws, wn = _lstm_ws(T, B, H, ndir, dev)
    else:                    # allocated and prefilled by the caller (lstm_prefill_fwd)
        wn = ws.numel()
        flags = 1            # DANET_LSTM_PREFILLED
    fused = (ldx % 4 == 0 and x.data_ptr() % 16 == 0 and (D % 4 == 0 or x_pad_zero) and
             all(W.stride(0) == 4 * H and W.stride(1) == 1 for W in Ws) and
             L.danet_lstm_fwd_fused_supported(T, B, H, ndir, D) == 1)
    if fused:
        # the whole cell [x_t, h_{t-1}] W + b (app/ops.py:139-147) in ONE launch: the input
        # half runs on the matrix cores inside the per-step exchange wait (csrc/lstm.hip)
        with _lib.timed('lstm_fwd'):
            check(L.danet_lstm_fwd_fused(
                _lib.stream(), T, B, H, ndir, ptr(_f32(x)), ldx, D,
                ptr(Ws[0]), ptr(Ws[-1]), 4 * H, ptr(bs[0]), ptr(bs[-1]), ptr(ypad), ndir * H,
                ptr(gates[0]), ptr(gates[-1]), ptr(cells[0]), ptr(cells[-1]), ptr(ws), wn,
                ptr(status_word(dev)), flags))
    else:
        if _x6_ok(T * B, 4 * H, (x, ldx, D)):
            # hoisted input half [x]Wx + b (app/ops.py:139-142) on the packed weight, one launch per
            # direction; `gates[d]` is later overwritten in place by g,i,f,o
            for d in range(ndir):
                gemm_w(x, ldx, Ws[d], 1, 4 * H, gates[d], T * B, 4 * H, D, 4 * H, tag='gx', bias=bs[d])
        elif GROUPED_GX and ndir == 2:
            # hoisted input half of ops.lyr_lstm_flat's [x,h]W+b (app/ops.py:139-142) of both
            # directions as one grouped stream-K launch (nothing else runs at this point of the
            # forward pass: 2 x 320 tiles fill 512 workgroups evenly; -1.5% per step vs two
            # launches on two streams); `gates[d]` is later overwritten in place by g,i,f,o
            gemm_group([(x, ldx, Ws[d], 4 * H, gates[d], 4 * H, T * B, 4 * H, 0.0, bs[d])
                        for d in range(ndir)], D, max_workgroups=GROUPED_GX)
        else:
            with _Fork(dev, ndir) as f:
                for d in range(ndir):
                    f.run(d, lambda d=d: gemm(x, Ws[d], gates[d], T * B, 4 * H, D, ldx, 4 * H,
                                              4 * H, bias=bs[d], tag='gx'))
        Whs = [W[D:] for W in Ws]
        with _lib.timed('lstm_fwd'):
            check(L.danet_lstm_fwd(
                _lib.stream(), T, B, H, ndir, ptr(gates[0]), ptr(gates[-1]),
                ptr(Whs[0]), ptr(Whs[-1]), 4 * H, ptr(ypad), ndir * H,
                ptr(gates[0]), ptr(gates[-1]), ptr(cells[0]), ptr(cells[-1]), ptr(ws), wn,
                ptr(status_word(dev)), flags))
    c = _LayerCtx()
    c.x, c.ldx, c.D, c.T, c.B, c.H, c.ndir = x, ldx, D, T, B, H, ndir
    c.ypad, c.gates, c.cells, c.Ws, c.bs = ypad, gates, cells, Ws, bs
    return c


# Do the weight-gradient groups run UNDER the next BPTT kernel (side stream) or serially on the
# main stream?  Under: the group is hidden but the BPTT kernel beside it slows down for as long as
# the overlap lasts.  Measured: H = 300 (cfg 2 / cfg 4) 3.61 / 5.13 ms per step overlapped vs
# 3.90 / 5.62 serial; H = 600 (cfg 4 as written) 14.3 overlapped vs 13.7 serial (the group lasts
# 0.9-1.3 ms there and costs the BPTT kernel 0.7 ms).  'auto' = overlap up to H = 384 with the
# exact-fp32 groups, always with the groups on the bf16 matrix cores (round 4).
DW_OVERLAP = _lib.expert('dw_overlap', 'auto')


def _overlap_dw(H, tn_ok=True):
    '''tn_ok: this group's products will actually take the bf16-matrix-core TN kernel
    (`_x6_tn_ok`); a group that falls back to the exact-fp32 stream-K kernels is overlapped only
    up to H = 384 (the combination measured slower above that)'''
    if DW_OVERLAP in ('0', '1'):
        return DW_OVERLAP == '1'
    # (with the groups on the bf16 matrix cores the overlap wins at H = 600 as well: cfg 4 as
    # written 9.66 overlapped vs 9.86 serial on the same box, BPTT 674 vs 575 us)
    return H <= 384 or (bool(GEMM_X6 & 2) and tn_ok)


# experiment: fork the weight-gradient group BEFORE dX (the two GEMMs share the GPU, the next
# BPTT kernel then has the group beside it for a shorter time)
DW_FORK_EARLY = _lib.expert('dw_fork_early', False)
# bias gradients summed inside the (unfused) BPTT kernel instead of by column-sum launches
BWD_DB = _lib.expert('lstm_bwd_db', True)
DB_DEFER = _lib.expert('lstm_db_defer', True)


def lstm_layer_bwd(c, dy, need_dx, layer_tag=None, is_top=False, ws_prefilled=None):
    '''dy: [T, B, ndir*H] contiguous.  Returns (dx [T*B, D] or None, dWs, dbs).
    ws_prefilled: a workspace whose ring the caller prefilled (lstm_prefill_train)'''
    T, B, H, D, ndir = c.T, c.B, c.H, c.D, c.ndir
    dev = dy.device
    das = [torch.empty(T * B, 4 * H, device=dev) for _ in range(ndir)]
    L = _L()
    ldy = ndir * H
    # gradients accumulate straight into the parameters' .grad (the model's flat
    # all-reduce bucket) when there is one; autograd then gets None for them
    dWs, dbs, direct = [], [], []
    for d in range(ndir):
        gW, okW = _grad_target(c.Ws[d], (D + H, 4 * H), dev)
        gb, okb = _grad_target(c.bs[d], (4 * H,), dev)
        dWs.append(gW); dbs.append(gb); direct.append((okW, okb))
    dx = torch.empty(T * B, D, device=dev) if need_dx else None
    if ws_prefilled is not None:
        ws, wn = ws_prefilled, ws_prefilled.numel()
    else:
        ws, wn = _lstm_ws(T, B, H, ndir, dev)
    Whs = [W[D:] for W in c.Ws]
    b_direct = [okb for _, okb in direct]
    # bias gradients summed inside the BPTT kernel (no column-sum launches) where the
    # reduce-scatter kernel covers the shape
    db_in_kernel = (BWD_DB and (all(b_direct) or not any(b_direct)) and
                    all(t.data_ptr() % 16 == 0 for t in dbs) and
                    L.danet_lstm_bwd_db_supported(T, B, H, ndir) == 1)
    # the 6-us sum of the kernel's per-cluster bias partials leaves the critical path (BPTT ->
    # dX -> next BPTT) when a side chain is forked behind this launch anyway
    # (not for the bottom layer: its weight-gradient group takes every CU on the main stream
    # and the side chain's reduce would sit behind it for the group's whole duration)
    def group_problems():
        # dWx = X^T da and dWh = Hprev^T da of every direction (Hprev(t) = ypad block t (fwd) / t+2 (bwd))
        probs = []
        for d in range(ndir):
            bW = 1.0 if direct[d][0] else 0.0
            hprev = c.ypad.view(-1)[(0 if d == 0 else 2 * B * ldy + H):]
            probs.append((c.x, c.ldx, das[d], 4 * H, dWs[d], 4 * H, D, 4 * H, bW))
            probs.append((hprev, ldy, das[d], 4 * H, dWs[d][D:], 4 * H, H, 4 * H, bW))
        return probs
    overlap = _overlap_dw(H, _x6_tn_ok(group_problems(), T * B))
    db_deferred = (db_in_kernel and DB_DEFER and GROUPED_DW and SIDE_STREAMS > 0 and
                   need_dx and overlap)
    with _lib.timed('lstm_bwd'):
        check(L.danet_lstm_bwd(
            _lib.stream(), T, B, H, ndir, ptr(_f32(dy)), ndir * H,
            ptr(Whs[0]), ptr(Whs[-1]), 4 * H, ptr(c.gates[0]), ptr(c.gates[-1]),
            ptr(c.cells[0]), ptr(c.cells[-1]), ptr(das[0]), ptr(das[-1]),
            ptr(dbs[0]) if db_in_kernel else None, ptr(dbs[-1]) if db_in_kernel else None,
            1.0 if all(b_direct) else 0.0, ptr(ws), wn, ptr(status_word(dev)),
            ((1 if ws_prefilled is not None else 0) | (2 if db_deferred else 0))
            if db_in_kernel else 0))

    # the weight-gradient products overlap the NEXT layer's BPTT kernel (152 of
    # 256 CUs at cfg 2): cap each chain so both together stay on the idle CUs
    cap = OVERLAP_GEMM_WGS if need_dx else 0

    def weight_grads(d):
        bW, bb = (1.0 if direct[d][0] else 0.0), (1.0 if direct[d][1] else 0.0)
        # dWx = X^T da
        gemm(c.x, das[d], dWs[d], D, 4 * H, T * B, c.ldx, 4 * H, 4 * H, transA=True, beta=bW,
             max_workgroups=cap)
        # dWh = Hprev^T da; Hprev(t) = ypad block t (fwd) / block t+2 (bwd)
        hprev = c.ypad.view(-1)[(0 if d == 0 else 2 * B * ldy + H):]
        gemm(hprev, das[d], dWs[d][D:], H, 4 * H, T * B, ldy, 4 * H, 4 * H, transA=True, beta=bW,
             max_workgroups=cap)
        if not db_in_kernel:
            colsum(das[d], T * B, 4 * H, 4 * H, dbs[d], beta=bb)

    def hprev_of(d):
        # Hprev(t) = ypad block t (fwd) / block t+2 (bwd)
        return c.ypad.view(-1)[(0 if d == 0 else 2 * B * ldy + H):]

    def weight_grads_grouped(wgs=GROUPED_DW_WGS, with_bias=True):
        # dWx = X^T da and dWh = Hprev^T da of every direction: one launch
        gemm_group(group_problems(), T * B, transA=True, max_workgroups=wgs)
        # (the bias gradients as M = 1 members of the group were measured slower than
        # the two column-sum kernels: +25 us on the group for 128-row tiles with one row)
        if with_bias:
            bias_grads()

    def bias_grads():
        if db_deferred:
            check(L.danet_lstm_bwd_db_reduce(_lib.stream(), T, B, H, ndir, ptr(dbs[0]), ptr(dbs[-1]),
                                             1.0 if direct[0][1] else 0.0, ptr(ws), wn))
            return
        if db_in_kernel:
            return
        for d in range(ndir):
            colsum(das[d], T * B, 4 * H, 4 * H, dbs[d], beta=1.0 if direct[d][1] else 0.0)

    fork_ev = [None]

    def input_grad(attach=False):
        if ndir == 2:
            # dX = da_f Wx_f^T + da_b Wx_b^T: one K-concatenated launch (the fork event of the
            # weight-gradient chain rides on it: see ForkEvent)
            sk = (STREAMK & 4) != 0 and (4 * H) % 16 == 0
            if _x6_ok(T * B, D, (das[0], 4 * H, 4 * H), (das[1], 4 * H, 4 * H)):
                if attach:
                    fork_ev[0] = fork_event(dev)
                gemm_w(das[0], 4 * H, c.Ws[0], 4 * H, 1, dx, T * B, D, 4 * H, D, tag='dX',
                       stop_event=fork_ev[0], A2=das[1], lda2=4 * H, W2=c.Ws[1], K2=4 * H)
                return
            if attach and sk:
                fork_ev[0] = fork_event(dev)
            gemm_kcat(das[0], 4 * H, c.Ws[0], 4 * H, 4 * H, das[1], 4 * H, c.Ws[1], 4 * H, 4 * H,
                      dx, T * B, D, D, transB=True, tag='dX', streamk=(STREAMK & 4) != 0,
                      stop_event=fork_ev[0])
            return
        for d in range(ndir):
            # dX += da Wx^T
            gemm(das[d], c.Ws[d], dx, T * B, D, 4 * H, 4 * H, 4 * H, D, transB=True,
                 beta=0.0 if d == 0 else 1.0, streamk=(STREAMK & 1) != 0, tag='dX')

    # dX is what the next layer's BPTT waits for: it is issued first, alone, on
    # the main stream.  The weight-gradient chains fork AFTER it (the fork event
    # is recorded behind dX), so they do not compete with dX for CUs but overlap
    # the next layer's latency-bound BPTT kernel instead; the caller joins them
    # (`join_deferred`).
    fork_early = DW_FORK_EARLY and need_dx and GROUPED_DW
    if need_dx and not fork_early:
        input_grad(attach=GROUPED_DW and overlap)
    hooks = bool(GRAD_READY_HOOKS) and _fast() and layer_tag is not None and \
        all(a and b for a, b in direct)
    if hooks and not need_dx:
        # bottom layer of an encoder: its BPTT kernel is on the main stream, the weight-
        # gradient chains of every layer above are on the side streams -- a stream that has
        # waited for all of them sees every gradient except this layer's in its final state
        sides = _side_streams(dev, max(SIDE_STREAMS, 1))
        main = torch.cuda.current_stream(dev)
        for sd in sides[1:]:
            sides[0].wait_stream(sd)
        sides[0].wait_stream(main)
        with torch.cuda.stream(sides[0]):
            _fire_grad_ready(('rest',), list(c.Ws) + list(c.bs))
    with _Fork(dev, ndir + 1, defer=True, keep=(das, c.x, c.ypad, dy, ws), lazy=True,
               event=fork_ev[0]) as f:
        on_main = False
        if GROUPED_DW and not need_dx:
            # bottom layer: no BPTT kernel follows, so the group takes the whole GPU on
            # the main stream while the column sums (if any) run beside it
            if not db_in_kernel:
                f.run(1, bias_grads)
            weight_grads_grouped(wgs=512, with_bias=False)
            on_main = True
        elif GROUPED_DW and not overlap:
            weight_grads_grouped(wgs=512)      # serial: alone on the main stream, whole GPU
            on_main = True
        elif GROUPED_DW:
            f.run(1, weight_grads_grouped)
            if fork_early:
                input_grad()          # beside the group, both behind this layer's BPTT kernel
        else:
            for d in range(ndir):
                f.run(d + 1, lambda d=d: weight_grads(d))
        if hooks:
            # (wait_main: the bottom layer's group GEMM ran on the main stream)
            f.after_all(lambda: _fire_grad_ready(('layer', layer_tag), list(c.Ws) + list(c.bs)),
                        wait_main=on_main)
    dWs = [None if direct[d][0] else dWs[d] for d in range(ndir)]
    dbs = [None if direct[d][1] else dbs[d] for d in range(ndir)]
    return dx, dWs, dbs


class LstmLayerFn(torch.autograd.Function):
    '''One (bi)LSTM layer on batch-major input: Model.lyr_lstm / _lyr_bilstm
    (main.py:76-132, app/modules.py:120-137).  x [B,T,D] -> [B,T,ndir*H]'''

    @staticmethod
    def forward(ctx, x, H, *params):
        ndir = len(params) // 2
        Ws, bs = list(params[0::2]), list(params[1::2])
        B, T, D = x.shape
        xt = x.transpose(0, 1).contiguous()         # tf.transpose, main.py:98-104
        c = lstm_layer_fwd(xt, D, D, T, B, H, Ws, bs)
        ctx.c = c
        ctx.xt = xt
        return c.ypad[1:T + 1].transpose(0, 1).contiguous()

    @staticmethod
    def backward(ctx, dy):
        c = ctx.c
        dyt = dy.transpose(0, 1).contiguous()
        dx, dWs, dbs = lstm_layer_bwd(c, dyt, ctx.needs_input_grad[0])
        join_deferred()
        out = [None, None]
        if dx is not None:
            out[0] = dx.view(c.T, c.B, c.D).transpose(0, 1).contiguous()
        for dW, db in zip(dWs, dbs):
            out += [dW, db]
        return tuple(out)


class RnnEncoderFn(torch.autograd.Function):
    '''Whole `bilstm-orig` / `lstm-orig` encoder (app/modules.py:148-260):
    mean-centre -> L stacked (bi)LSTM layers -> mean-centre -> bias-free output
    projection.  x [B,T,F] -> embed [B,T,F*E].
    params = (W_0f, b_0f, [W_0b, b_0b], W_1f, ..., W_out)'''

    @staticmethod
    def forward(ctx, x, H, L, ndir, *params):
        x = _f32(x.contiguous())
        B, T, F = x.shape
        dev = x.device
        Wout = params[-1]
        Fp = (F + 3) // 4 * 4
        # x - mean_{t,f}(x), switched to time-major, zero-padded to a float4 row       modules.py:209-210
        xc = torch.empty(T, B, Fp, device=dev)
        ctxs = []
        cur, ld, D = xc, Fp, F
        # output buffers + workspaces of ALL layers (a train step: + the BPTT launches' rings), prefilled
        # by ONE fill -- which rides in the centring launch (danet_encoder_prologue)
        ypads = [torch.empty(T + 2, B, ndir * H, device=dev) for _ in range(L)]
        wss = [_lstm_ws(T, B, H, ndir, dev)[0] for _ in range(L)]
        bwss = None
        if any(ctx.needs_input_grad) and BWD_DB and _L().danet_lstm_bwd_db_supported(T, B, H, ndir) == 1:
            bwss = [_lstm_ws(T, B, H, ndir, dev)[0] for _ in range(L)]
        if not ENC_PROLOGUE:
            center(x, B, T, F, 0, F, xc, 1, Fp)
            if bwss is None or not lstm_prefill_train(T, B, H, ndir, ypads, wss, bwss):
                bwss = None
                lstm_prefill_fwd(T, B, ndir * H, ypads, wss)
        else:
            if not encoder_prologue(x, B, T, F, xc, Fp, H, ndir, ypads, wss, bwss):
                bwss = None
        ctx.bwss = bwss
        for l in range(L):                                    # modules.py:223-242
            Ws = [params[(l * ndir + d) * 2] for d in range(ndir)]
            bs = [params[(l * ndir + d) * 2 + 1] for d in range(ndir)]
            # (the centre kernel zero-fills the float4 pad of layer 0's rows; deeper layers
            # read ypad, whose row length is a multiple of 4)
            c = lstm_layer_fwd(cur, ld, D, T, B, H, Ws, bs, x_pad_zero=True, ypad=ypads[l],
                               ws=wss[l])
            ctxs.append(c)
            cur, ld, D = c.ypad[1:], ndir * H, ndir * H
        # y - mean_{t,h}(y), back to batch-major                modules.py:244-245
        yc = torch.empty(B, T, D, device=dev)
        center(cur, B, T, D, 1, D, yc, 0, D)
        O = Wout.shape[1]
        embed = torch.empty(B, T, O, device=dev)
        # (hybrid stream-K pays for the projection only with >= 4 tiles per CU: cfg 4, 1312 tiles,
        # 248 -> 238 us; cfg 2, 672 tiles, 126 -> 132 us)
        big = ((B * T + 127) // 128) * ((O + 127) // 128) >= 1024
        if _x6_ok(B * T, O, (yc, D, D)):
            gemm_w(yc, D, Wout, 1, O, embed, B * T, O, D, O, tag='proj')     # modules.py:249-255
        else:
            gemm(yc, Wout, embed, B * T, O, D, D, O, O, tag='proj',
                 streamk=(STREAMK & 2) != 0 or (big and STREAMK != 0))
        ctx.ctxs, ctx.yc, ctx.Wout = ctxs, yc, Wout
        ctx.dims = (B, T, F, H, L, ndir, D, O)
        return embed

    @staticmethod
    def backward(ctx, dembed):
        B, T, F, H, L, ndir, D, O = ctx.dims
        dembed = _f32(dembed.contiguous())
        dev = dembed.device
        dWout, direct_out = _grad_target(ctx.Wout, (D, O), dev)
        dyc = torch.empty(B, T, D, device=dev)
        x6 = _x6_ok(B * T, D, (dembed, O, O))
        ev = fork_event(dev) if ((STREAMK & 1) != 0 or x6) else None
        if x6:                                                # critical path first
            gemm_w(dembed, O, ctx.Wout, O, 1, dyc, B * T, D, O, D, tag='dYc', stop_event=ev)
        else:
            gemm(dembed, ctx.Wout, dyc, B * T, D, O, O, O, D, transB=True, streamk=(STREAMK & 1) != 0,
                 tag='dYc', stop_event=ev)
        with _Fork(dev, 2, defer=True, keep=(dembed, ctx.yc), lazy=True, event=ev) as f:
            if GROUPED_DW:    # alone on its stream under the top layer's BPTT kernel (or serial)
                ov = _overlap_dw(H, _x6_tn_ok([(ctx.yc, D, dembed, O, dWout, O, D, O, 0.0)], B * T))
                f.run(1 if ov else 0, lambda: gemm_group(
                    [(ctx.yc, D, dembed, O, dWout, O, D, O, 1.0 if direct_out else 0.0)], B * T,
                    transA=True, max_workgroups=GROUPED_DW_WGS if ov else 512))
            else:
                f.run(1, lambda: gemm(ctx.yc, dembed, dWout, D, O, B * T, D, O, O, transA=True,
                                      beta=1.0 if direct_out else 0.0,
                                      max_workgroups=2 * OVERLAP_GEMM_WGS, tag='dWout'))
            if GRAD_READY_HOOKS and _fast() and direct_out:
                f.after_all(lambda: _fire_grad_ready(('out',), [ctx.Wout]))
        dy = torch.empty(T, B, D, device=dev)
        center(dyc, B, T, D, 0, D, dy, 1, D)                 # centre is self-adjoint
        grads = [None] * (2 * L * ndir)
        # partial-dh rings of all layers' BPTT launches: prefilled by the forward pass's fill launch
        bwss = ctx.bwss if ctx.bwss is not None else [None] * L
        ctx.bwss = None
        for l in reversed(range(L)):
            dx, dWs, dbs = lstm_layer_bwd(ctx.ctxs[l], dy, need_dx=(l > 0), layer_tag=l,
                                          is_top=(l == L - 1), ws_prefilled=bwss[l])
            for d in range(ndir):
                grads[(l * ndir + d) * 2] = dWs[d]
                grads[(l * ndir + d) * 2 + 1] = dbs[d]
            dy = dx
        join_deferred()
        ctx.ctxs = None
        return (None, None, None, None) + tuple(grads) + (None if direct_out else dWout,)


class LinearFn(torch.autograd.Function):
    '''ops.lyr_linear on the last axis (app/ops.py:72-78,81-89)'''

    @staticmethod
    def forward(ctx, x, W, b):
        shp = x.shape
        x2 = _f32(x.contiguous()).view(-1, shp[-1])
        M, K = x2.shape
        N = W.shape[1]
        y = torch.empty(M, N, device=x.device)
        gemm(x2, W, y, M, N, K, K, N, N, bias=b)
        ctx.save_for_backward(x2, W)
        ctx.has_b = b is not None
        ctx.shp = shp
        return y.view(*shp[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        x2, W = ctx.saved_tensors
        M, K = x2.shape
        N = W.shape[1]
        dy2 = _f32(dy.contiguous()).view(M, N)
        dW = torch.empty(K, N, device=dy.device)
        gemm(x2, dy2, dW, K, N, M, K, N, N, transA=True)
        db = None
        if ctx.has_b:
            db = torch.empty(N, device=dy.device)
            colsum(dy2, M, N, N, db)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, device=dy.device)
            gemm(dy2, W, dx, M, K, N, N, N, K, transB=True)
            dx = dx.view(ctx.shp)
        return dx, dW, db


def lyr_linear(x, W, b=None):
    return LinearFn.apply(x, W, b)


class LeakyReluFn(torch.autograd.Function):
    '''ops.relu (app/ops.py:93-107) as a HIP kernel pair'''

    @staticmethod
    def forward(ctx, x, alpha):
        x = _f32(x.contiguous())
        y = torch.empty_like(x)
        check(_L().danet_leaky_relu(_lib.stream(), x.numel(), ptr(x), None, float(alpha), ptr(y)))
        ctx.save_for_backward(x)
        ctx.alpha = float(alpha)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        dx = torch.empty_like(x)
        check(_L().danet_leaky_relu(_lib.stream(), x.numel(), ptr(x), ptr(_f32(dy.contiguous())),
                                    ctx.alpha, ptr(dx)))
        return dx, None


def relu(s_x, alpha=0.):
    '''app/ops.py:93-107 (toy encoder)'''
    return LeakyReluFn.apply(s_x, alpha)


# ---------------------------------------------------------------------------
# estimators / separators / loss
# ---------------------------------------------------------------------------
TRUTH_MODES = {'truth': 0, 'truth-threshold': 1, 'truth-weighted': 2}


# The embedding feeds both the estimator and the separator; autograd would materialise
# the two gradient contributions (42 MB each at cfg 2) and add them.  The separator's
# backward runs first (it consumes the attractors): it leaves its dembed here and the
# estimator's backward -- whose kernels accumulate (`dembed += ...`) -- adds into that
# buffer in place and returns no gradient of its own.  DANET_FUSE_DEMBED=0 disables.
FUSE_DEMBED = _lib.expert('fuse_dembed', 1)
# Inside Model.train_step (heads chain) with the anchor estimator: the fused separator + loss
# backward produces only dattr and the estimator's backward forms the WHOLE embedding gradient in
# one pass (danet_attractor_anchor_bwd_embed_sep) -- the separator's term is not written to HBM
# and read back.  DANET_HEADS_RECOMPUTE=0: the two-pass accumulate-in-place form.
HEADS_RECOMPUTE = _lib.expert('heads_recompute', 1)
# ... and, where the library offers it (C == 2), the fused separator + loss FORWARD already leaves the
# attractor-gradient partials of every permutation behind: its backward then launches nothing at all -- the
# estimator's backward derives the permutation from the records, adds up that permutation's partials and
# uses them as dattr (round 6: one read of the embedding, danet_separate_pit_bwd and its chunk sum gone from
# the train step).  DANET_EXPERT heads_gradfwd=0: the round-5 form.
HEADS_GRADFWD = _lib.expert('heads_gradfwd', 1)


# "Heads chain" scope (entered by Model.train_step around forward + backward): inside it the two
# reductions nobody on the critical path waits for are issued on the side stream --
#   * the loss / SNR / permutation finalize of SeparatePitFn (the backward kernel derives the
#     permutation from the chunk records itself), and
#   * the anchors' gradient finalize of AnchorAttractorFn (only the optimiser reads it)
# -- and joined by `join_deferred()` (RnnEncoderFn.backward ends with it; train_step calls it
# again before the optimiser).  Outside the scope both run in stream order, so a caller that
# reads `loss` right after forward needs no join.
_chain_depth = [0]


class heads_chain(object):
    def __enter__(self):
        _chain_depth[0] += 1
        return self

    def __exit__(self, *exc):
        _chain_depth[0] -= 1
        return False


def _chain():
    return _chain_depth[0] > 0 and SIDE_STREAMS > 0


_lazy = {}          # device index -> [(fn, keep)]


def _dev_key(dev):
    if dev is not None and dev.type != 'cuda':
        return dev.type                    # (host tensors: the data-parallel CPU tests)
    return dev.index if (dev is not None and dev.index is not None) else torch.cuda.current_device()


def _on_side(dev, fn, keep=()):
    '''fn() runs on the side stream at the NEXT fork of the backward pass (behind that fork's
    event, i.e. behind everything issued on the main stream so far), ahead of the fork's own
    work; the main stream joins it at the next join_deferred().  No fork of its own: recording
    an event on the main stream costs it ~8 us (the following kernel cannot overlap the
    previous one's tail), more than the small kernels moved here take.  Without any fork before
    the join, fn() runs in stream order at the join.'''
    _lazy.setdefault(_dev_key(dev), []).append((fn, keep))


def _flush_lazy(dev=None):
    '''run the queued closures of `dev` (default: the current device) on the current stream; a
    closure that raises leaves nothing queued behind it'''
    q = _lazy.get(_dev_key(dev))
    try:
        while q:
            fn, _keep = q.pop(0)
            fn()
    finally:
        if q:
            del q[:]


def drop_lazy(dev=None):
    '''forget the queued side-stream closures of `dev` (Model.train_step: forward raised, the
    finalizers of that step must not run inside the next step's first fork)'''
    q = _lazy.get(_dev_key(dev))
    if q:
        del q[:]


class _DembedToken(object):
    '''links an estimator's autograd node to the separator node that consumes its
    attractors: created in the estimator's forward, carried on the attractor tensor
    (`attr._danet_dembed_token`; lost -- and the fusion with it -- if the caller
    transforms the attractors in between), picked up by SeparateFn.forward.  No global
    state, so several models / re-entrant backward passes cannot alias each other.'''
    __slots__ = ('dembed', 'kind', 'recipe', 'inputs')

    def __init__(self):
        self.dembed = None
        self.kind = None          # 'anchor': the estimator's backward can recompute the separator's term
        self.recipe = None        # set by SeparatePitFn.backward when it left that term to the estimator
        self.inputs = None        # (embed ptr, numel, mix_pwr ptr or None) the estimator saw

    def same_inputs(self, embed_flat, mix_pwr):
        '''the recompute kernels form the separator's term from the ESTIMATOR's saved embedding
        (and, for the truth family, are handed the separator's mix_pwr where the estimator's own
        is expected): only valid when both modules were given the same tensors'''
        if self.inputs is None:
            return False
        e_ptr, e_n, m_ptr = self.inputs
        return (embed_flat.data_ptr() == e_ptr and embed_flat.numel() == e_n and
                (m_ptr is None or mix_pwr.data_ptr() == m_ptr))


def _new_token(attr):
    if not FUSE_DEMBED:
        return None
    tok = _DembedToken()
    attr._danet_dembed_token = tok
    return tok


def _take_dembed(tok, numel):
    if tok is None:
        return None
    t, tok.dembed = tok.dembed, None
    if t is not None and t.numel() == numel:
        return t
    return None


class TruthAttractorFn(torch.autograd.Function):
    '''app/modules.py:382-487'''

    @staticmethod
    def forward(ctx, embed, src_pwr, mix_pwr, mode, eps):
        B, T, F, E = embed.shape
        C = src_pwr.shape[1]
        N = T * F
        dev = embed.device
        embed = _f32(embed.contiguous())
        src_pwr = _f32(src_pwr.contiguous())
        mix_pwr = _f32(mix_pwr.contiguous())
        attr = torch.empty(B, C, E, device=dev)
        denom = torch.empty(B, C, device=dev)
        L = _L()
        w, wn = _ws(_lib.ws_bytes(_lib.WS_ATTRACTOR_TRUTH, B, C, N, E), dev)
        check(L.danet_attractor_truth_fwd(_lib.stream(), mode, B, C, N, E, ptr(embed),
                                          ptr(src_pwr), ptr(mix_pwr), eps, ptr(attr),
                                          ptr(denom), ptr(w), wn))
        ctx.save_for_backward(src_pwr, mix_pwr, denom, embed, attr)
        ctx.args = (mode, eps, B, C, N, E, T, F)
        ctx.token = _new_token(attr)
        if ctx.token is not None:
            ctx.token.kind = 'truth'
            ctx.token.inputs = (embed.data_ptr(), embed.numel(), mix_pwr.data_ptr())
        return attr

    @staticmethod
    def backward(ctx, dattr):
        src_pwr, mix_pwr, denom, embed, attr = ctx.saved_tensors
        mode, eps, B, C, N, E, T, F = ctx.args
        recipe = None
        if ctx.token is not None:
            recipe, ctx.token.recipe = ctx.token.recipe, None
        if recipe is not None:
            # the separator's embedding-gradient term is recomputed here (one pass, one store)
            act, lmode, s_mix, src, phasor, records, dl, gpart = recipe
            dembed = torch.empty(B, T, F, E, device=dattr.device)
            # (gpart: dattr is formed by the kernel from the forward's partials; the tensor autograd
            # handed over is a placeholder)
            check(_L().danet_attractor_truth_bwd_sep(
                _lib.stream(), mode, B, C, N, E, None if gpart is not None else ptr(_f32(dattr.contiguous())),
                ptr(src_pwr), ptr(s_mix), ptr(denom), eps, ptr(embed), ptr(attr), act, lmode,
                ptr(torch.view_as_real(src)), ptr(phasor), None, ptr(records), 1.0, ptr(dl),
                ptr(dembed), ptr(gpart)))
            return dembed, None, None, None, None
        shared = _take_dembed(ctx.token, B * T * F * E)
        dembed = shared if shared is not None else torch.zeros(B, T, F, E, device=dattr.device)
        check(_L().danet_attractor_truth_bwd(
            _lib.stream(), mode, B, C, N, E, ptr(_f32(dattr.contiguous())), ptr(src_pwr),
            ptr(mix_pwr), ptr(denom), eps, ptr(dembed)))
        return (None if shared is not None else dembed), None, None, None, None


class AnchorAttractorFn(torch.autograd.Function):
    '''app/modules.py:490-545.  Returns (attr, asets, subset_choice)'''

    @staticmethod
    def forward(ctx, embed, anchors, C):
        B, T, F, E = embed.shape
        A = anchors.shape[0]
        N = T * F
        dev = embed.device
        embed = _f32(embed.contiguous())
        anchors_in = anchors
        anchors = _f32(anchors.contiguous())
        P = math.comb(A, C)
        attr = torch.empty(B, C, E, device=dev)
        asets = torch.empty(B, P, C, E, device=dev)
        asum = torch.empty(B, P, C, device=dev)
        choice = torch.empty(B, dtype=torch.int32, device=dev)
        L = _L()
        w, wn = _ws(_lib.ws_bytes(_lib.WS_ATTRACTOR_ANCHOR, B, C, N, E, A), dev)
        check(L.danet_attractor_anchor_fwd(_lib.stream(), B, C, N, E, A, ptr(embed),
                                           ptr(anchors), ptr(attr), ptr(asets), ptr(asum),
                                           ptr(choice), ptr(w), wn))
        ctx.save_for_backward(embed, anchors, attr, asum, choice)
        ctx.anchors_param = anchors_in      # the parameter object (owner of .grad), not a copy
        ctx.args = (B, C, N, E, A, T, F)
        ctx.mark_non_differentiable(asets, choice)
        ctx.set_materialize_grads(False)
        ctx.token = _new_token(attr)
        if ctx.token is not None:
            ctx.token.kind = 'anchor'
            ctx.token.inputs = (embed.data_ptr(), embed.numel(), None)
        return attr, asets, choice

    @staticmethod
    def backward(ctx, dattr, _dasets, _dchoice):
        if dattr is None:
            return None, None, None
        embed, anchors, attr, asum, choice = ctx.saved_tensors
        B, C, N, E, A, T, F = ctx.args
        dev = dattr.device
        recipe = None
        if ctx.token is not None:
            recipe, ctx.token.recipe = ctx.token.recipe, None
        shared = None if recipe is not None else _take_dembed(ctx.token, B * T * F * E)
        if recipe is not None:
            dembed = torch.empty(B, T, F, E, device=dev)       # written whole by the kernel
        else:
            dembed = shared if shared is not None else torch.zeros(B, T, F, E, device=dev)
        # fast backward: add straight into the parameter's .grad (no autograd accumulate kernel)
        danchors, direct = _grad_target(ctx.anchors_param, (A, E), dev)
        L = _L()
        nbytes = _lib.ws_bytes(_lib.WS_ATTRACTOR_ANCHOR, B, C, N, E, A)
        dattr = _f32(dattr.contiguous())
        if recipe is not None:
            # the separator's embedding-gradient term is recomputed here (one pass, one store)
            act, mode, mix_pwr, src, phasor, records, dl, gpart = recipe
            w = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            check(L.danet_attractor_anchor_bwd_embed_sep(
                _lib.stream(), B, C, N, E, A, None if gpart is not None else ptr(dattr), ptr(embed),
                ptr(anchors), ptr(attr), ptr(asum), ptr(choice), act, mode, ptr(mix_pwr),
                ptr(torch.view_as_real(src)), ptr(phasor), None, ptr(records), 1.0, ptr(dl), ptr(dembed),
                ptr(w), nbytes, ptr(gpart)))
            _on_side(dev, lambda: check(L.danet_attractor_anchor_bwd_anchors(
                _lib.stream(), B, C, N, E, A, ptr(choice), ptr(danchors), ptr(w), nbytes,
                1.0 if direct else 0.0)), keep=(w, choice, danchors))
        elif _chain():
            # part 1 (dembed) on this stream; part 2 (danchors, from the chunk partials) on the
            # side stream: its own scratch, kept alive until the join
            w = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            check(L.danet_attractor_anchor_bwd_embed(
                _lib.stream(), B, C, N, E, A, ptr(dattr), ptr(embed), ptr(anchors), ptr(attr),
                ptr(asum), ptr(choice), ptr(dembed), ptr(w), nbytes))
            _on_side(dev, lambda: check(L.danet_attractor_anchor_bwd_anchors(
                _lib.stream(), B, C, N, E, A, ptr(choice), ptr(danchors), ptr(w), nbytes,
                1.0 if direct else 0.0)), keep=(w, choice, danchors))
        else:
            w, wn = _ws(nbytes, dev)
            check(L.danet_attractor_anchor_bwd_embed(
                _lib.stream(), B, C, N, E, A, ptr(dattr), ptr(embed), ptr(anchors), ptr(attr),
                ptr(asum), ptr(choice), ptr(dembed), ptr(w), wn))
            check(L.danet_attractor_anchor_bwd_anchors(
                _lib.stream(), B, C, N, E, A, ptr(choice), ptr(danchors), ptr(w), wn,
                1.0 if direct else 0.0))
        return (None if shared is not None else dembed), (None if direct else danchors), None


class SeparateFn(torch.autograd.Function):
    '''app/modules.py:548-603.  act 0 softmax / 1 sigmoid.
    (mix_pwr [B,T,F], attr [B,C,E], embed_flat [B,N,E]) -> (sep [B,C,T,F], masks)'''

    @staticmethod
    def forward(ctx, mix_pwr, attr, embed_flat, act, want_masks):
        B, T, F = mix_pwr.shape
        C, E = attr.shape[1], attr.shape[2]
        N = T * F
        dev = mix_pwr.device
        mix_pwr = _f32(mix_pwr.contiguous())
        attr = _f32(attr.contiguous())
        embed_flat = _f32(embed_flat.contiguous())
        out = torch.empty(B, C, T, F, device=dev)
        masks = torch.empty(B, T, F, C, device=dev) if want_masks else None
        check(_L().danet_separate_fwd(_lib.stream(), act, B, C, N, E, ptr(mix_pwr), ptr(attr),
                                      ptr(embed_flat), ptr(out), ptr(masks)))
        ctx.save_for_backward(mix_pwr, attr, embed_flat)
        ctx.args = (act, B, C, N, E)
        ctx.token = getattr(attr, '_danet_dembed_token', None)
        if masks is None:
            masks = torch.empty(0, device=dev)
        ctx.mark_non_differentiable(masks)
        ctx.set_materialize_grads(False)
        return out, masks

    @staticmethod
    def backward(ctx, dout, _dmasks):
        if dout is None:
            return None, None, None, None, None
        mix_pwr, attr, embed_flat = ctx.saved_tensors
        act, B, C, N, E = ctx.args
        dev = dout.device
        dembed = torch.empty(B, N, E, device=dev)
        dattr = torch.empty(B, C, E, device=dev)
        L = _L()
        w, wn = _ws(_lib.ws_bytes(_lib.WS_SEPARATE_BWD, B, C, N, E), dev)
        check(L.danet_separate_bwd(_lib.stream(), act, B, C, N, E, ptr(mix_pwr), ptr(attr),
                                   ptr(embed_flat), ptr(_f32(dout.contiguous())), ptr(dembed),
                                   ptr(dattr), ptr(w), wn))
        if ctx.token is not None and ctx.needs_input_grad[1]:
            # the estimator that made `attr` runs its backward next and adds into `dembed`
            ctx.token.dembed = dembed
        return None, dattr, dembed, None, None


class PitMseFn(torch.autograd.Function):
    '''ops.pit_mse_loss (app/ops.py:374-431) fused with the phase re-attach
    (main.py:281-284) and batch_snr (app/ops.py:191-222).
    mode 0: complex MSE (train, main.py:289-290); mode 1: magnitude MSE
    (valid, main.py:312-313).  Returns (loss, snr, perm_idx)'''

    @staticmethod
    def forward(ctx, src, sep_pwr, phasor, mode, eps):
        assert src.dtype == torch.complex64
        B, C, T, F = sep_pwr.shape
        N = T * F
        dev = sep_pwr.device
        src = src.contiguous()
        sep_pwr = _f32(sep_pwr.contiguous())
        phasor = _f32(phasor.contiguous())
        loss, snr = torch.empty((), device=dev), torch.empty((), device=dev)
        perm_idx = torch.empty(B, dtype=torch.int32, device=dev)
        L = _L()
        w, wn = _ws(_lib.ws_bytes(_lib.WS_PIT_MSE, B, C, N), dev)
        check(L.danet_pit_mse_fwd(_lib.stream(), mode, B, C, N, ptr(torch.view_as_real(src)),
                                  ptr(sep_pwr), ptr(phasor), eps, ptr(loss), ptr(snr),
                                  ptr(perm_idx), ptr(w), wn))
        ctx.save_for_backward(src, sep_pwr, phasor, perm_idx)
        ctx.args = (mode, B, C, N)
        ctx.mark_non_differentiable(snr, perm_idx)
        ctx.set_materialize_grads(False)
        return loss, snr, perm_idx

    @staticmethod
    def backward(ctx, dloss, _dsnr, _dperm):
        src, sep_pwr, phasor, perm_idx = ctx.saved_tensors
        mode, B, C, N = ctx.args
        if dloss is None:
            return None, None, None, None, None
        dsep = torch.empty_like(sep_pwr)
        # dloss is a device scalar: the kernel reads it (no host sync, no extra pass)
        check(_L().danet_pit_mse_bwd(_lib.stream(), mode, B, C, N,
                                     ptr(torch.view_as_real(src)), ptr(sep_pwr), ptr(phasor),
                                     ptr(perm_idx), 1.0, ptr(_f32(dloss.contiguous())), ptr(dsep)))
        return None, dsep, None, None, None


class SeparatePitFn(torch.autograd.Function):
    '''Separator + phase re-attach + PIT-MSE + SNR in ONE pass over the embedding and ONE
    pass back (danet_separate_pit_fwd_records + _final / _bwd): the training path of app/modules.py:548-603 ->
    main.py:281-290, 308-309 -> app/ops.py:374-431.  Same arithmetic as SeparateFn followed by
    PitMseFn; the masks, the separated magnitudes and dL/dsep never touch HBM.
    (mix_pwr [B,T,F], attr [B,C,E], embed_flat [B,N,E], src complex64 [B,C,T,F],
    phasor [B,T,F,2]) -> (loss, snr, perm_idx)'''

    @staticmethod
    def forward(ctx, mix_pwr, attr, embed_flat, src, phasor, act, mode, eps):
        assert src.dtype == torch.complex64
        B, T, F = mix_pwr.shape
        C, E = attr.shape[1], attr.shape[2]
        N = T * F
        dev = mix_pwr.device
        mix_pwr = _f32(mix_pwr.contiguous())
        attr_c = _f32(attr.contiguous())
        embed_flat = _f32(embed_flat.contiguous())
        src = src.contiguous()
        phasor = _f32(phasor.contiguous())
        loss, snr = torch.empty((), device=dev), torch.empty((), device=dev)
        perm_idx = torch.empty(B, dtype=torch.int32, device=dev)
        L = _L()
        records = torch.empty(_lib.ws_bytes(_lib.WS_SEPARATE_PIT_RECORDS, B, N) // 4, device=dev)
        # the backward's attractor-gradient partials already here?  Only where the backward will be able
        # to leave everything to the estimator's backward (the conditions of `defer` below, as far as
        # they are known now) and the library offers it for the shape
        tok = getattr(attr, '_danet_dembed_token', None)
        gpart = None
        if (HEADS_GRADFWD and HEADS_RECOMPUTE and _chain() and tok is not None and
                tok.kind in ('anchor', 'truth') and ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and
                tok.same_inputs(embed_flat, mix_pwr)):
            nb = _lib.ws_bytes(_lib.WS_SEPARATE_PIT_GRAD, B, C, N, E)
            if nb:
                gpart = torch.empty(nb // 4, device=dev)
        check(L.danet_separate_pit_fwd_records(
            _lib.stream(), act, mode, B, C, N, E, ptr(mix_pwr), ptr(attr_c), ptr(embed_flat),
            ptr(torch.view_as_real(src)), ptr(phasor), None, ptr(records), ptr(gpart)))
        ctx.gpart = gpart

        def final():
            check(L.danet_separate_pit_final(_lib.stream(), B, C, N, eps, ptr(records), ptr(loss),
                                             ptr(snr), ptr(perm_idx)))
        if _chain():
            _on_side(dev, final, keep=(records, loss, snr, perm_idx))   # the backward does not wait for it
        else:
            final()
        ctx.records = records
        ctx.save_for_backward(mix_pwr, attr_c, embed_flat, src, phasor, perm_idx)
        ctx.args = (act, mode, B, C, N, E)
        ctx.token = getattr(attr, '_danet_dembed_token', None)
        ctx.mark_non_differentiable(snr, perm_idx)
        ctx.set_materialize_grads(False)
        return loss, snr, perm_idx

    @staticmethod
    def backward(ctx, dloss, _dsnr, _dperm):
        if dloss is None:
            return (None,) * 8
        mix_pwr, attr, embed_flat, src, phasor, perm_idx = ctx.saved_tensors
        act, mode, B, C, N, E = ctx.args
        records, ctx.records = ctx.records, None
        gpart, ctx.gpart = ctx.gpart, None
        dev = dloss.device
        tok = ctx.token
        defer = (HEADS_RECOMPUTE and _chain() and tok is not None and tok.kind in ('anchor', 'truth') and
                 ctx.needs_input_grad[1] and ctx.needs_input_grad[2] and
                 tok.same_inputs(embed_flat, mix_pwr))
        dembed = None if defer else torch.empty(B, N, E, device=dev)
        dattr = torch.empty(B, C, E, device=dev)
        dl = _f32(dloss.contiguous())
        L = _L()
        if defer and gpart is not None:
            # nothing to launch: the estimator's backward (next node) forms dattr from the forward's
            # partials and the whole embedding gradient; `dattr` is a placeholder that only keeps
            # autograd walking to that node
            tok.recipe = (act, mode, mix_pwr, src, phasor, records, dl, gpart)
            return None, dattr, None, None, None, None, None, None
        w, wn = _ws(_lib.ws_bytes(_lib.WS_SEPARATE_PIT, B, C, N, E), dev)
        # dloss is a device scalar: the kernel reads it (no host sync, no extra pass)
        check(L.danet_separate_pit_bwd(_lib.stream(), act, mode, B, C, N, E, ptr(mix_pwr),
                                       ptr(attr), ptr(embed_flat), ptr(torch.view_as_real(src)),
                                       ptr(phasor), None, ptr(records), 1.0,
                                       ptr(dl), ptr(dembed), ptr(dattr), ptr(w), wn))
        if defer:
            # the estimator's backward (next node) recomputes this kernel's dembed term
            # and returns the whole embedding gradient
            tok.recipe = (act, mode, mix_pwr, src, phasor, records, dl, None)
            return None, dattr, None, None, None, None, None, None
        if ctx.token is not None and ctx.needs_input_grad[1]:
            # the estimator that made `attr` runs its backward next and adds into `dembed`
            ctx.token.dembed = dembed
        return None, dattr, dembed, None, None, None, None, None


def separate_pit_loss(s_mix_pwr, s_attractors, s_embed_flat, s_x, phasor, act, mode=0, eps=1e-7):
    '''fused separator + pit_mse_loss; returns (s_loss, v_perms, s_loss_sets_idx, snr) like
    pit_mse_loss'''
    C = s_attractors.shape[1]
    loss, snr, idx = SeparatePitFn.apply(s_mix_pwr, s_attractors, s_embed_flat, s_x, phasor,
                                         act, mode, eps)
    v_perms = torch.tensor(list(itertools.permutations(range(C))), dtype=torch.int32)
    return loss, v_perms, idx, snr


def pit_mse_loss(s_x, s_y_pwr, phasor, mode=0, eps=1e-7):
    '''returns (s_loss, v_perms, s_loss_sets_idx, snr) like app/ops.py:374-431'''
    C = s_y_pwr.shape[1]
    loss, snr, idx = PitMseFn.apply(s_x, s_y_pwr, phasor, mode, eps)
    v_perms = torch.tensor(list(itertools.permutations(range(C))), dtype=torch.int32)
    return loss, v_perms, idx, snr


def adam_clip_step(theta, grad, m, v, lr_t, beta1=0.9, beta2=0.999, eps=1e-8, clip=100.0,
                   grad_scale=1.0, zero_grad=False):
    '''flat fp32 buffers; tf.train.AdamOptimizer + clip_by_value
    (main.py:359-363, app/ozers.py:15-18)'''
    check(_L().danet_adam_clip_step(_lib.stream(), theta.numel(), ptr(_f32(theta)),
                                    ptr(_f32(grad)), ptr(_f32(m)), ptr(_f32(v)), lr_t, beta1,
                                    beta2, eps, clip if clip else 0.0, grad_scale,
                                    int(bool(zero_grad))))
    weights_written(theta)
