'''
ctypes binding of libdanet_hip.so (the C ABI declared in include/danet_hip.h).

There is NO fallback: if the shared library is missing or a call fails, a
RuntimeError is raised.  PyTorch is used only to own device memory and streams;
tensors cross the boundary as raw device pointers.
'''
import ctypes
import os
import threading

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# DANET_LIB_PATH: an A/B build of the same sources (_build.build_variant), never a different backend
LIB_PATH = os.environ.get('DANET_LIB_PATH') or os.path.join(_HERE, 'csrc', 'libdanet_hip.so')

c_int, c_i64, c_f32, c_sz, c_p = (ctypes.c_int, ctypes.c_int64, ctypes.c_float,
                                  ctypes.c_size_t, ctypes.c_void_p)

class GemmPack(ctypes.Structure):
    '''danet_gemm_pack_t (include/danet_hip.h)'''
    _fields_ = [('src', c_p), ('stride_n', ctypes.c_longlong), ('stride_k', ctypes.c_longlong),
                ('N', c_int), ('K', c_int), ('out', c_p), ('out_bytes', c_sz)]


class GemmProblem(ctypes.Structure):
    '''danet_gemm_problem_t (include/danet_hip.h)'''
    _fields_ = [('A', c_p), ('lda', c_int), ('B', c_p), ('ldb', c_int), ('C', c_p), ('ldc', c_int),
                ('M', c_int), ('N', c_int), ('bias', c_p), ('beta', c_f32)]


# name -> (restype, argtypes); mirrors include/danet_hip.h
PROTOTYPES = {
    'danet_abi_version': (c_int, []),
    'danet_last_error': (ctypes.c_char_p, []),
    'danet_set_option': (c_int, [ctypes.c_char_p, c_int]),
    'danet_get_option': (c_int, [ctypes.c_char_p, ctypes.POINTER(c_int)]),
    'danet_reset_options': (None, []),
    'danet_option_count': (c_int, []),
    'danet_option_name': (ctypes.c_char_p, [c_int]),
    'danet_workspace_bytes': (c_sz, [c_int, ctypes.POINTER(c_i64), c_int]),
    'danet_stft_num_frames': (c_int, [c_i64, c_int, c_int]),
    'danet_stft': (c_int, [c_p, c_int, c_i64, c_int, c_int, c_p, c_p, c_p]),
    'danet_istft': (c_int, [c_p, c_int, c_int, c_int, c_int, c_p, c_p, c_p, c_p, c_sz]),
    'danet_frontend_fwd': (c_int, [c_p, c_int, c_int, c_i64, c_p, c_p, c_p, c_p, c_p, c_p, c_p]),
    'danet_reattach_phase': (c_int, [c_p, c_int, c_int, c_i64, c_p, c_p, c_p, c_p]),
    'danet_center': (c_int, [c_p, c_int, c_int, c_int, c_p, c_int, c_int, c_p, c_int, c_int, c_p]),
    'danet_gemm_f32': (c_int, [c_p, c_int, c_int, c_int, c_int, c_int, c_p, c_int, c_p, c_int,
                               c_p, c_int, c_p, c_f32, c_p, c_sz, c_int]),
    'danet_gemm_f32_streamk': (c_int, [c_p, c_int, c_int, c_int, c_int, c_int, c_p, c_int, c_p, c_int,
                                       c_p, c_int, c_p, c_f32, c_p, c_sz]),
    'danet_gemm_f32_streamk_grouped': (c_int, [c_p, c_int, c_int, c_int, c_int,
                                               ctypes.POINTER(GemmProblem), c_int, c_p, c_sz]),
    'danet_gemm_f32_streamk_kcat': (c_int, [c_p, c_int, c_int, c_int, c_int,
                                            c_int, c_p, c_int, c_p, c_int, c_int, c_p, c_int, c_p, c_int,
                                            c_p, c_int, c_p, c_f32, c_p, c_sz]),
    'danet_gemm_pack_weights': (c_int, [c_p, c_int, ctypes.POINTER(GemmPack)]),
    'danet_gemm_x6': (c_int, [c_p, c_int, c_int, c_int, c_p, c_int, c_p, c_int, c_p, c_int, c_p,
                              c_p, c_int, c_p, c_p, c_sz]),
    'danet_gemm_x6_tn_grouped': (c_int, [c_p, c_int, c_int, ctypes.POINTER(GemmProblem), c_p, c_sz]),
    'danet_colsum_f32': (c_int, [c_p, c_int, c_int, c_p, c_int, c_p, c_f32, c_p, c_sz]),
    'danet_lstm_fwd': (c_int, [c_p, c_int, c_int, c_int, c_int, c_p, c_p, c_p, c_p, c_int,
                               c_p, c_int, c_p, c_p, c_p, c_p, c_p, c_sz, c_p, c_int]),
    'danet_lstm_fwd_prefill': (c_int, [c_p, c_int, c_int, c_int, c_int, ctypes.POINTER(c_p),
                                       ctypes.POINTER(c_p)]),
    'danet_encoder_prologue': (c_int, [c_p, c_int, c_int, c_int, c_p, c_int, c_int, c_p, c_int, c_int, c_p,
                                       c_int, c_int, c_int, c_int, ctypes.POINTER(c_p), ctypes.POINTER(c_p),
                                       ctypes.POINTER(c_p)]),
    'danet_lstm_train_prefill': (c_int, [c_p, c_int, c_int, c_int, c_int, c_int, c_int, ctypes.POINTER(c_p),
                                         ctypes.POINTER(c_p), ctypes.POINTER(c_p)]),
    'danet_lstm_fwd_fused_supported': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'danet_lstm_fwd_fused_supported': (c_int, [c_int, c_int, c_int, c_int, c_int]),
    'danet_lstm_fwd_fused': (c_int, [c_p, c_int, c_int, c_int, c_int, c_p, c_int, c_int, c_p, c_p,
                                     c_int, c_p, c_p, c_p, c_int, c_p, c_p, c_p, c_p, c_p, c_sz, c_p,
                                     c_int]),
    'danet_lstm_bwd': (c_int, [c_p, c_int, c_int, c_int, c_int, c_p, c_int, c_p, c_p, c_int,
                               c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_p, c_f32, c_p, c_sz, c_p, c_int]),
    'danet_lstm_bwd_db_supported': (c_int, [c_int, c_int, c_int, c_int]),
    'danet_lstm_bwd_db_reduce': (c_int, [c_p, c_int, c_int, c_int, c_int, c_p, c_p, c_f32, c_p, c_sz]),
    'danet_next_launch_events': (c_int, [c_p, c_p]),
    'danet_event_create': (c_int, [ctypes.POINTER(c_p)]),
    'danet_event_destroy': (c_int, [c_p]),
    'danet_stream_wait_event': (c_int, [c_p, c_p]),
    'danet_attractor_truth_fwd': (c_int, [c_p, c_int, c_int, c_int, c_i64, c_int, c_p, c_p, c_p,
                                          c_f32, c_p, c_p, c_p, c_sz]),
    'danet_attractor_truth_bwd': (c_int, [c_p, c_int, c_int, c_int, c_i64, c_int, c_p, c_p, c_p,
                                          c_p, c_f32, c_p]),
    'danet_attractor_truth_bwd_sep': (c_int, [c_p, c_int, c_int, c_int, c_i64, c_int, c_p, c_p, c_p, c_p,
                                              c_f32, c_p, c_p, c_int, c_int, c_p, c_p, c_p, c_p, c_f32,
                                              c_p, c_p, c_p]),
    'danet_attractor_anchor_fwd': (c_int, [c_p, c_int, c_int, c_i64, c_int, c_int, c_p, c_p, c_p,
                                           c_p, c_p, c_p, c_p, c_sz]),
    'danet_separate_fwd': (c_int, [c_p, c_int, c_int, c_int, c_i64, c_int, c_p, c_p, c_p, c_p, c_p]),
    'danet_separate_bwd': (c_int, [c_p, c_int, c_int, c_int, c_i64, c_int, c_p, c_p, c_p, c_p,
                                   c_p, c_p, c_p, c_sz]),
    'danet_separate_pit_bwd': (c_int, [c_p, c_int, c_int, c_int, c_int, c_i64, c_int, c_p, c_p, c_p,
                                       c_p, c_p, c_p, c_p, c_f32, c_p, c_p, c_p, c_p, c_sz]),
    'danet_separate_pit_bwd': (c_int, [c_p, c_int, c_int, c_int, c_int, c_i64, c_int, c_p, c_p, c_p,
                                       c_p, c_p, c_p, c_p, c_f32, c_p, c_p, c_p, c_p, c_sz]),
    'danet_separate_pit_fwd_records': (c_int, [c_p, c_int, c_int, c_int, c_int, c_i64, c_int, c_p, c_p,
                                               c_p, c_p, c_p, c_p, c_p, c_p]),
    'danet_separate_pit_final': (c_int, [c_p, c_int, c_int, c_i64, c_f32, c_p, c_p, c_p, c_p]),
    'danet_attractor_anchor_bwd_embed': (c_int, [c_p, c_int, c_int, c_i64, c_int, c_int, c_p, c_p, c_p,
                                                 c_p, c_p, c_p, c_p, c_p, c_sz]),
    'danet_attractor_anchor_bwd_embed_sep': (c_int, [c_p, c_int, c_int, c_i64, c_int, c_int, c_p, c_p, c_p,
                                                     c_p, c_p, c_p, c_int, c_int, c_p, c_p, c_p, c_p, c_p,
                                                     c_f32, c_p, c_p, c_p, c_sz, c_p]),
    'danet_attractor_anchor_bwd_anchors': (c_int, [c_p, c_int, c_int, c_i64, c_int, c_int, c_p, c_p,
                                                   c_p, c_sz, c_f32]),
    'danet_pit_mse_fwd': (c_int, [c_p, c_int, c_int, c_int, c_i64, c_p, c_p, c_p, c_f32, c_p,
                                  c_p, c_p, c_p, c_sz]),
    'danet_pit_mse_bwd': (c_int, [c_p, c_int, c_int, c_int, c_i64, c_p, c_p, c_p, c_p, c_f32, c_p, c_p]),
    'danet_leaky_relu': (c_int, [c_p, c_i64, c_p, c_p, c_f32, c_p]),
    'danet_adam_clip_step': (c_int, [c_p, c_i64, c_p, c_p, c_p, c_p, c_f32, c_f32, c_f32, c_f32,
                                     c_f32, c_f32, c_int]),
}

_lib = None
_lock = threading.Lock()


class DanetHipError(RuntimeError):
    pass


def load():
    '''dlopen libdanet_hip.so (after torch, so both share ONE libamdhip64).'''
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise DanetHipError(
                'libdanet_hip.so not found at %s -- the HIP extension is required '
                '(there is no CPU fallback); build it with '
                '`python -c "import __graft_entry__ as g; g.build()"`' % LIB_PATH)
        # torch already mapped its bundled libamdhip64.so (SONAME libamdhip64.so.7);
        # our DT_NEEDED of the same SONAME resolves to that copy.
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_LOCAL)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(lib, name)      # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if lib.danet_abi_version() != 7:
            raise DanetHipError('libdanet_hip.so ABI version mismatch')
        _lib = lib
        apply_env_options()
    return _lib


# ---- switches ------------------------------------------------------------------
# USER switches (README): DANET_GEMM_X6, DANET_LSTM_FWD_FUSED, DANET_SIDE_STREAMS, DANET_FEED_MODE,
# DANET_OVERLAP_ALLREDUCE, DANET_ALLREDUCE_TAIL_RATIO, DANET_MAX_STEPS_IN_FLIGHT, DANET_STATUS_HOST, DANET_FUSE_HEADS,
# DANET_LSTM_SPIN_LIMIT, DANET_LSTM_FAULT_INJECT, DANET_LIB_PATH (+ three of bench.py).  Everything
# else -- the schedule knobs of the exact-fp32 fallback kernels, placement and fork details, the
# remaining library options -- is an EXPERT setting behind ONE variable:
#     DANET_EXPERT="streamk=5,grouped_dw=0,gemm_yield=8"
# (names: the lower-case module constants of ops.py / model.py that call `expert()`, and the library
# options of csrc/options.h).  Defaults are the shipped, measured configuration.
_expert = None
_expert_seen = set()      # every name some module has asked for (typos in DANET_EXPERT are reported, below)


def expert(name, default):
    '''value of the expert setting `name` (DANET_EXPERT="name=value,...") or `default`, as
    type(default)'''
    global _expert
    if _expert is None:
        _expert = {}
        for item in os.environ.get('DANET_EXPERT', '').split(','):
            if item.strip():
                k, _, v = item.partition('=')
                _expert[k.strip().lower()] = v.strip()
    _expert_seen.add(name.lower())
    v = _expert.get(name.lower())
    if v is None:
        return default
    if isinstance(default, bool):
        return v not in ('0', 'false', 'no', '')
    return type(default)(v)


USER_OPTIONS = ('lstm_fwd_fused', 'lstm_spin_limit', 'lstm_fault_inject')   # library options with a DANET_<NAME> variable


# ---- library options ---------------------------------------------------------
# The C library never reads the environment (include/danet_hip.h); the DANET_* overrides of
# its options live HERE: option `lstm_fwd_fused` <- env DANET_LSTM_FWD_FUSED, applied when the
# library is loaded (and by apply_env_options(), which tests call to restore the defaults).
def option_names():
    lib = _lib
    return [lib.danet_option_name(i).decode() for i in range(lib.danet_option_count())]


def apply_env_options():
    '''reset every option to its default, then apply DANET_<NAME> from the environment'''
    lib = _lib
    lib.danet_reset_options()
    for name in option_names():
        v = os.environ.get('DANET_' + name.upper()) if name in USER_OPTIONS else None
        if v is None or v.strip() == '':
            v = expert(name, '')
        if v != '':
            check(lib.danet_set_option(name.encode(), int(v)))
    # a key of DANET_EXPERT that neither a module constant nor a library option answers to is a typo: say so
    # (the library is loaded at the first kernel call, after ops.py / model.py have read their settings)
    unknown = sorted(set(_expert or {}) - _expert_seen)
    if unknown:
        import warnings
        warnings.warn('DANET_EXPERT: unknown setting(s) %s (known: the lower-case constants of ops.py / model.py '
                      'that call _lib.expert(), and the library options %s)' % (unknown, option_names()))


def set_option(name, value):
    check(load().danet_set_option(name.encode(), int(value)))


def get_option(name):
    v = c_int(0)
    check(load().danet_get_option(name.encode(), ctypes.byref(v)))
    return v.value


def check(rc):
    if rc != 0:
        msg = load().danet_last_error()
        raise DanetHipError('libdanet_hip error %d: %s' % (
            rc, msg.decode() if msg else '?'))


# DANET_WS_* (include/danet_hip.h)
(WS_ISTFT, WS_GEMM, WS_GEMM_STREAMK, WS_COLSUM, WS_LSTM, WS_ATTRACTOR_TRUTH, WS_ATTRACTOR_ANCHOR,
 WS_SEPARATE_BWD, WS_SEPARATE_PIT, WS_SEPARATE_PIT_RECORDS, WS_PIT_MSE, WS_CENTER_MEAN,
 WS_GEMM_X6, WS_GEMM_PACK, WS_GEMM_X6_TN, WS_SEPARATE_PIT_GRAD) = range(16)


def ws_bytes(op, *dims):
    '''danet_workspace_bytes(op, dims): scratch bytes of the entry point behind DANET_WS_<op>'''
    arr = (c_i64 * len(dims))(*[int(d) for d in dims])
    n = load().danet_workspace_bytes(op, arr, len(dims))
    if n == ctypes.c_size_t(-1).value:
        check(-1)
    return n


def ptr(t):
    '''raw device pointer of a torch tensor (None -> NULL)'''
    if t is None:
        return None
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


# ---- scratch ---------------------------------------------------------------
_ws = {}


def workspace(nbytes, device, tag=None):
    '''per-(device, stream, tag) grow-only scratch, zero-initialised; safe because
    every library call is stream-ordered on the current stream and finishes with
    `ws` before the next call on that stream starts.'''
    key = (device.index if device.index is not None else torch.cuda.current_device(),
           torch.cuda.current_stream().cuda_stream, tag)
    t = _ws.get(key)
    if t is None or t.numel() < nbytes:
        t = torch.zeros(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
        _ws[key] = t
    return t


# ---- optional per-kernel timing (HIP events on the launch stream) ------------
_prof = None
_prof_only = None
_prof_on = True


def profile_start(only=None):
    '''only: optional set of labels to time (every event pair costs ~1 us of stream
    time, so a timed benchmark region instruments just the kernel it reports)'''
    global _prof, _prof_only
    if torch.cuda.is_available():
        prepare_timing(64)          # (a no-op after prepare_timing())
    _prof = {}
    _prof_only = set(only) if only else None


def profile_enable(flag):
    '''pause / resume event recording inside a profile_start()..profile_stop() window'''
    global _prof_on
    _prof_on = bool(flag)


def profile_stop():
    '''-> {label: (n_launches, total_ms)}; synchronises'''
    global _prof
    p, _prof = _prof, None
    torch.cuda.synchronize()
    out = {}
    for label, evs in (p or {}).items():
        out[label] = (len(evs), sum(a.elapsed_time(b) for a, b in evs))
    return out


# labels whose ONE kernel launch carries the event pair on its own dispatch packet
# (danet_next_launch_events) instead of two event records around the call: the recurrent kernels.
# A record in front of the launch delays it by a few microseconds, enough for the side stream's
# weight-gradient group to get its workgroups onto the CUs first -- the persistent kernel then
# starts piecemeal and the bracket reads 50 us more than the kernel takes in an untimed step.
ATTACHED_LABELS = frozenset(('lstm_fwd', 'lstm_bwd'))
_aux_stream = None
_event_pool = []


def _timing_event():
    '''a timing-enabled event whose native handle exists (torch creates it on first record; that
    record goes to a stream nobody else uses).  profile_start() fills a pool, so that a timed
    region creates neither a stream nor events (a stream created inside a 20-step region cost it
    5-6 ms on some boxes).'''
    global _aux_stream
    if _event_pool:
        return _event_pool.pop()
    return _timing_event_new()


def prepare_timing(n=384):
    '''create the side stream and a pool of timing events NOW (call before the warm-up steps of a
    benchmark: creating them, and the synchronisation behind it, must not sit at the start of a
    timed region)'''
    made = False
    while len(_event_pool) < n:
        _event_pool.append(_timing_event_new())
        made = True
    if made:
        torch.cuda.synchronize()


def _timing_event_new():
    global _aux_stream
    if _aux_stream is None:
        _aux_stream = torch.cuda.Stream()
    e = torch.cuda.Event(enable_timing=True)
    e.record(_aux_stream)
    return e


class timed(object):
    '''with timed('label'): <one library call>  -- records a start/end event pair
    on the current stream (the stream the kernels are launched on) when
    profiling is enabled; free otherwise.  Labels in ATTACHED_LABELS: the pair rides on the
    call's own kernel launch.'''
    __slots__ = ('label', 'tag', 'a', 'b')

    def __init__(self, label, tag=None):
        self.label, self.tag = label, tag

    def __enter__(self):
        self.a = self.b = None
        if _prof is not None and _prof_on and (_prof_only is None or self.label in _prof_only):
            if self.label in ATTACHED_LABELS:
                self.a, self.b = _timing_event(), _timing_event()
                check(load().danet_next_launch_events(self.a.cuda_event, self.b.cuda_event))
            else:
                self.a = torch.cuda.Event(enable_timing=True)
                self.a.record()
        return self

    def __exit__(self, *exc):
        if self.a is not None and _prof is not None:
            b = self.b
            if b is None:
                b = torch.cuda.Event(enable_timing=True)
                b.record()
            _prof.setdefault(self.label, []).append((self.a, b))
            if self.tag:      # per-call-site breakdown next to the per-entry-point total
                _prof.setdefault(self.label + ':' + self.tag, []).append((self.a, b))
        return False
