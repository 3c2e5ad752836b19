'''
Model: the forward graph of the reference's `Model.build()` (main.py:208-399)
re-stated eagerly over the HIP ops, plus the train / valid / infer steps the
reference runs through `g_sess.run` (main.py:402-532, :685-690).

Host code is Python + PyTorch-ROCm (autograd engine, device memory, streams,
torch.distributed); every tensor op on the path is a libdanet_hip.so kernel.

Data parallelism (new; the reference is single-GPU, README.md:226): one process
per GPU, `hparams.BATCH_SIZE` is the per-GPU batch, parameters live in one flat
fp32 buffer broadcast from rank 0, gradients accumulate into one flat fp32
bucket that is all-reduced ONCE per step over RCCL, then
value-clipped (main.py:359-362) and applied by the TF1-style Adam kernel.
'''
import math
import os

import numpy as np
import torch

from .hparams import hparams
from . import _lib, ops
from . import modules  # noqa: F401  (registers the plugins)
from . import ozers    # noqa: F401  (registers the optimisers)
from . import dist


class Model(object):
    '''Base class for a fully trainable model (main.py:61-548)'''

    def __init__(self, name='BaseModel', device=None, seed=1337, grad_schedule=None):
        '''grad_schedule (data parallelism only; default: env DANET_OVERLAP_ALLREDUCE or 'auto'):
        '0' = ONE all-reduce of the flat gradient bucket after backward (the north_star
        form); 'tail' = two collectives (everything outside the bottom encoder layer under
        that layer's weight-gradient GEMMs, the rest after backward); '1' = per-layer buckets
        under the remaining BPTT kernels (opt-in only).  'auto' (default) starts as '0' and
        DECIDES between '0' and 'tail' after AUTO_DECIDE_AT train steps from two measurements of
        its own -- the stand-alone all-reduce time of its bucket and its step time, MAX-reduced
        over the ranks -- with one threshold (dist.choose_schedule, DANET_ALLREDUCE_TAIL_RATIO);
        `schedule_decision` records what was measured.  Without a process group 'auto' is '0'.'''
        self.grad_schedule = str(grad_schedule if grad_schedule is not None else
                                 os.environ.get('DANET_OVERLAP_ALLREDUCE', 'auto'))
        assert self.grad_schedule in ('0', 'tail', '1', 'auto'), self.grad_schedule
        self.schedule_decision = None
        self._auto = self.grad_schedule == 'auto'
        if self._auto:
            self.grad_schedule = '0'
        self.name = name
        self.device = torch.device(device if device is not None else
                                   'cuda:%d' % torch.cuda.current_device())
        self.s_states_di = {}
        self.vars = {}
        self._order = []
        self._gen = torch.Generator().manual_seed(seed)
        self._flat = None
        self.learn_rate = float(hparams.LR)
        self.step_count = 0
        self.built = False
        # True: gradients stay readable after train_step (grad_dict); costs one fill
        # kernel per step.  False: the optimiser kernel zeroes them after use.
        self.keep_grads = False
        # train_step: separator + PIT loss as one fused kernel pair (DANET_FUSE_HEADS=0: off)
        self.fuse_heads = os.environ.get('DANET_FUSE_HEADS', '1') == '1'

    # ------------------------------------------------------------ variables
    def get_variable(self, name, shape, init):
        '''tf.get_variable under scope "global/" (main.py:229).  `init(shape,
        generator)` returns a CPU float32 tensor.'''
        full = 'global/' + name
        v = self.vars.get(full)
        if v is None:
            if self._flat is not None:
                raise KeyError('variable %s requested after build()' % full)
            v = init(list(shape), self._gen).to(self.device).requires_grad_(True)
            assert list(v.shape) == list(shape), (full, v.shape, shape)
            self.vars[full] = v
            self._order.append(full)
        return v

    def lyr_lstm(self, name, s_x, hdim, axis=-1, t_axis=0, op_linear=None,
                 w_init=None, b_init=None):
        '''single unidirectional LSTM layer (main.py:76-132); s_x [B,T,D]
        (t_axis=-2/1) or [T,B,D] (t_axis=0) -> same layout with D -> hdim.
        Zero initial state; state variables are never carried across calls
        (main.py:108-123, :538-540).'''
        assert s_x.dim() == 3 and axis in (-1, 2)
        t_axis = t_axis % 3
        assert t_axis in (0, 1)
        if t_axis == 0:
            s_x = s_x.transpose(0, 1)
        D = s_x.shape[-1]
        W = self.get_variable(name + '/LSTM/linear/W', [D + hdim, 4 * hdim], w_init)
        b = self.get_variable(name + '/LSTM/linear/B', [4 * hdim], b_init)
        y = ops.LstmLayerFn.apply(s_x, hdim, W, b)
        return y.transpose(0, 1) if t_axis == 0 else y

    def parameter_count(self):
        return sum(v.numel() for v in self.vars.values())

    # ---------------------------------------------------------------- build
    def build(self):
        '''create sub-modules (main.py:210-211, 249-250, 263-270), materialise
        variables with one dry forward, then flatten them for the optimiser.'''
        self.encoder = hparams.get_encoder()(self, 'encoder')
        self.estimator = hparams.get_estimator(
            hparams.TRAIN_ESTIMATOR_METHOD)(self, 'train_estimator')
        self.using_same_method = (
            hparams.INFER_ESTIMATOR_METHOD == hparams.TRAIN_ESTIMATOR_METHOD)
        if self.using_same_method:
            self.valid_estimator = self.estimator
        else:
            self.valid_estimator = hparams.get_estimator(
                hparams.INFER_ESTIMATOR_METHOD)(self, 'infer_estimator')
            assert not self.valid_estimator.USE_TRUTH           # main.py:266
        self.separator = hparams.get_separator(hparams.SEPARATOR_TYPE)(self, 'separator')
        B, C, F = hparams.BATCH_SIZE, hparams.MAX_N_SIGNAL, hparams.FEATURE_SIZE
        dry = torch.zeros(B, C, 4, F, dtype=torch.complex64, device=self.device)
        with torch.no_grad():
            self.forward(dry, with_valid=True)
        self._flatten()
        self.ozer = hparams.get_optimizer()(
            learn_rate=self.learn_rate, lr_decay=hparams.LR_DECAY)
        self.ozer.bind(self._flat, self._flat_grad)
        self.built = True
        dist.broadcast_params_(self._flat)
        ops.weights_written(self._flat)
        return self

    def _flatten(self):
        n = sum(self.vars[k].numel() for k in self._order)
        flat = torch.empty(n, device=self.device)
        # 4 spare floats IN FRONT of the gradients: under data parallelism they carry the
        # persistent kernels' hand-off status through the gradient all-reduce (ops.py).  In front,
        # because the variables are laid out bottom encoder layer first: the LAST collective of
        # every schedule ('0': the whole bucket; 'tail': the bottom layer's range, reduced after
        # backward) then covers them as part of one contiguous range -- behind every kernel of the
        # step, wherever a schedule launches its early pieces (round 6; behind the gradients they
        # travelled with 'tail''s early piece, safe only as long as that piece starts after the
        # last recurrent kernel)
        self._grad_store = torch.zeros(n + 4, device=self.device)
        grad = self._grad_store[4:]
        off = 0
        for k in self._order:
            v = self.vars[k]
            m = v.numel()
            flat[off:off + m].copy_(v.detach().reshape(-1))
            nv = flat[off:off + m].view(v.shape).requires_grad_(True)
            nv.grad = grad[off:off + m].view(v.shape)
            self.vars[k] = nv
            off += m
        self._flat, self._flat_grad = flat, grad
        ops.drop_packs(self.device)       # (the dry run packed weights at their old addresses)
        self._one = torch.ones((), device=self.device)
        # a backward pass outside train_step (tests, user code) leaves gradients behind:
        # autograd's accumulation marks the bucket dirty so the next train_step clears it
        self._grads_clean = True
        import weakref
        me = weakref.ref(self)

        def _mark_dirty(_p):
            m = me()
            if m is not None:
                m._grads_clean = False
        for k in self._order:
            self.vars[k].register_post_accumulate_grad_hook(_mark_dirty)
        # gradient reduction schedule (dist.py, see __init__): '0' = one all-reduce after backward;
        # 'tail' / '1' = overlapped pieces; 'auto' starts as '0' and decides after a few steps
        self._buckets = None
        offs, off = {}, 0
        for k in self._order:
            v = self.vars[k]
            offs[v.data_ptr()] = (off, off + v.numel())
            off += v.numel()
        self._offs = offs
        self._install_schedule(self.grad_schedule)
        if self._auto and not dist.is_dist():
            self._auto = False            # nothing to decide without a process group
            self.schedule_decision = dict(schedule='0', reason='single process')
        self._auto_ev = None
        # early optimizer step (opt-in: DANET_EXPERT early_adam=1): once the bottom encoder layer's BPTT
        # kernel has been issued every other gradient is final (and, under data parallelism with
        # the 'tail' schedule, reduced), so clip + Adam over everything outside that layer's range
        # can run on the side stream under the bottom layer's kernels; after backward only the
        # bottom layer's range is left.  The update is elementwise and nothing reads those
        # parameters again in this step (bit-identical parameters, tested).  Round 2 measured
        # -35 us per cfg-2 step; with round 3's shorter tail it LOSES 27 us (3.166 vs 3.192 ms:
        # the 188 MB Adam stream beside the bottom layer's weight-gradient group costs the group
        # more than the 33 us the single update takes afterwards) -> off by default.
        self._early_hooked = False
        self._early_adam = _lib.expert('early_adam', False)
        if dist.is_dist():
            # the status word rides in the gradient all-reduce: every rank sees the same value
            ops.set_status_word(self.device, self._grad_store[:4].view(torch.int32))
        elif ops.status_word(self.device).is_cuda and ops.STATUS_HOST:
            ops.set_status_word(self.device, None)     # a previous data-parallel model's word
        self._early = None          # (ranges, stream) of this step's early update
        self.early_steps = 0        # steps that took the early path (diagnostics / tests)
        self._in_step = False

    # ---- 'auto': '0' or 'tail', decided from the model's own measurements ------------------------
    AUTO_DECIDE_AT = 6       # train steps before the decision; steps 4..6 are the timed ones

    def _install_schedule(self, mode):
        self.grad_schedule = mode
        if mode in ('1', 'tail') and self._buckets is None:
            cls = dist.GradBuckets if mode == '1' else dist.TailOverlap
            if dist.is_dist():      # ranges inside the store: 4 status words in front, reduced LAST
                offs = {k: (lo + 4, hi + 4) for k, (lo, hi) in self._offs.items()}
                self._buckets = cls(self._grad_store, offs, last=[(0, 4)])
            else:
                self._buckets = cls(self._flat_grad, self._offs)
            ops.add_grad_ready_hook(self._buckets.hook)

    def _auto_tick(self):
        '''called at the end of every train step while the schedule is undecided.  Every rank runs
        the same steps, so every rank reaches the decision -- which contains collectives -- at the
        same point; the inputs are MAX-reduced, so every rank decides the same.'''
        n = self.step_count
        if n == self.AUTO_DECIDE_AT - 3:
            self._auto_ev = torch.cuda.Event(enable_timing=True)
            self._auto_ev.record()
        if n < self.AUTO_DECIDE_AT or self._auto_ev is None:
            return
        end = torch.cuda.Event(enable_timing=True)
        end.record()
        end.synchronize()
        step_ms = dist.allreduce_max_scalar(self._auto_ev.elapsed_time(end) / 3.0, self.device)
        ar_ms = dist.measure_allreduce_ms(self._grad_store.numel(), self.device)
        mode = dist.choose_schedule(ar_ms, step_ms)
        self.schedule_decision = dict(schedule=mode, allreduce_ms=round(ar_ms, 4), step_ms=round(step_ms, 4),
                                      ratio=round(ar_ms / step_ms, 5), threshold=dist.TAIL_RATIO,
                                      bucket_bytes=4 * self._grad_store.numel(), decided_at_step=n,
                                      world=dist.world_size())
        self._auto, self._auto_ev = False, None
        self._install_schedule(mode)

    # (the gradient-ready hook is registered only when the early step is wanted: with no hook at
    # all ops does not fire the 'rest' event, whose cross-stream wait costs the main stream ~10 us)
    @property
    def _early_adam(self):
        return self._early_adam_on

    @_early_adam.setter
    def _early_adam(self, on):
        self._early_adam_on = bool(on)
        if on and not self._early_hooked:
            ops.add_grad_ready_hook(self._grad_ready)     # after the bucket's hook: it launches first
            self._early_hooked = True

    def _grad_ready(self, tag, params):
        if tag[0] != 'rest' or not self._early_adam or not self._in_step or self._early is not None:
            return
        rng = [self._offs.get(p.data_ptr()) for p in params]
        if any(r is None for r in rng):
            return                      # not this model's encoder
        if dist.is_dist():
            if not isinstance(self._buckets, dist.TailOverlap):
                return                  # the other schedules reduce these gradients later
            self._buckets.wait_launched()
        ranges = dist._complement(rng, self._flat_grad.numel())
        self.ozer.step(self.step_count + 1, self.learn_rate, clip=hparams.GRAD_CLIP_THRES,
                       grad_scale=1.0 / dist.world_size(), zero_grad=not self.keep_grads,
                       ranges=ranges)
        ops.repack_weights(self.device)
        self._early = (ranges, torch.cuda.current_stream(self.device))
        self.early_steps += 1

    # -------------------------------------------------------------- forward
    def forward(self, s_src_signals, with_valid=False, with_train=True, fuse_heads=False):
        '''main.py:215-337.  s_src_signals complex64 [B, C, T, F].
        fuse_heads (train_step): separator + phase re-attach + PIT loss + SNR run as ONE
        kernel forward and ONE backward (ops.SeparatePitFn); the separated magnitudes then
        exist only in registers and `sep_pwr` is not in the returned dict.  Needs a separator
        that exposes its activation (`ACT`, the dot-product separators); others take the
        unfused path.'''
        B, E = hparams.BATCH_SIZE, hparams.EMBED_SIZE
        eps = float(hparams.EPS)
        fe = ops.frontend(s_src_signals)                       # main.py:233-240
        s_embed = self.encoder(fe['mix_log'])                  # main.py:243
        s_embed_flat = s_embed.reshape(B, -1, E)               # main.py:244-246
        out = dict(embed=s_embed, input=s_src_signals)
        phasor = fe['phasor']
        if with_train:
            s_attractors = self.estimator(                     # main.py:251-254
                s_embed, s_src_pwr=fe['src_pwr'], s_mix_pwr=fe['mix_pwr'])
            act = getattr(self.separator, 'ACT', None)
            if fuse_heads and act is not None and not hparams.DEBUG and not with_valid:
                loss, perms, idx, snr = ops.separate_pit_loss(      # :271-272 + :281-290, 308-309
                    fe['mix_pwr'], s_attractors, s_embed_flat, s_src_signals, phasor, act,
                    mode=0, eps=eps)
                out.update(attrs=s_attractors, loss=loss, SNR=snr, perm_idx=idx, perms=perms)
            else:
                s_sep_pwr = self.separator(fe['mix_pwr'], s_attractors, s_embed_flat)  # :271-272
                loss, perms, idx, snr = ops.pit_mse_loss(          # main.py:289-290, 308-309
                    s_src_signals, s_sep_pwr, phasor, mode=0, eps=eps)
                out.update(attrs=s_attractors, sep_pwr=s_sep_pwr, loss=loss, SNR=snr,
                           perm_idx=idx, perms=perms)
        if with_valid:
            if self.using_same_method and with_train:
                s_vattr, s_vsep = out['attrs'], out['sep_pwr']
            else:
                if self.valid_estimator.USE_TRUTH:
                    s_vattr = self.valid_estimator(
                        s_embed, s_src_pwr=fe['src_pwr'], s_mix_pwr=fe['mix_pwr'])
                else:
                    # main.py:267 passes only the embedding; the mixture magnitude is an
                    # extra keyword the reference's estimator signature already has
                    # (app/modules.py:501), used by the k-means extension as weights
                    s_vattr = self.valid_estimator(s_embed, s_mix_pwr=fe['mix_pwr'])
                s_vsep = self.separator(fe['mix_pwr'], s_vattr, s_embed_flat)  # :277-278
            vloss, perms, vidx, vsnr = ops.pit_mse_loss(       # main.py:312-313, 336-337
                s_src_signals, s_vsep, phasor, mode=1, eps=eps)
            out.update(valid_attrs=s_vattr, sep_pwr_valid=s_vsep, valid_loss=vloss,
                       valid_SNR=vsnr, valid_perm_idx=vidx, perms=perms)
        out['phasor'] = phasor
        out['mix_pwr'] = fe['mix_pwr']
        return out

    def debug_fetch(self, s_src_signals):
        '''the `-m debug` fetch list (main.py:387-397): embed, attrs, input,
        output (+ module debug_fetches when hparams.DEBUG)'''
        with torch.no_grad():
            o = self.forward(s_src_signals)
            res = dict(embed=o['embed'], attrs=o['attrs'], input=s_src_signals,
                       output=ops.reattach_phase(o['sep_pwr'], o['phasor'], o['perm_idx']))
            for mod in (self.encoder, self.separator, self.estimator):
                res.update(getattr(mod, 'debug_fetches', {}) or {})
        return res

    # ------------------------------------------------------------ train step
    def train_step(self, s_src_signals, sync_metrics=True):
        '''one `g_sess.run(train_fetches)` (main.py:430-431): forward, backward,
        gradient all-reduce, value clip, Adam.  Returns dict(loss, SNR, LR) of
        device scalars (no host sync unless the caller reads them).'''
        # bounded run-ahead + status of the steps that have completed (raises DanetHipError
        # at most ops.MAX_STEPS_IN_FLIGHT steps after a hand-off timeout)
        ops.poll_status(self.device)
        if not self._grads_clean:
            self._flat_grad.zero_()
        chain = ops.heads_chain()        # small finalize kernels go to the side stream (ops.py)
        chain.__enter__()
        try:
            out = self.forward(s_src_signals, fuse_heads=self.fuse_heads)
        except BaseException:
            chain.__exit__(None, None, None)
            ops.drop_lazy(self.device)     # this step's queued finalizers must not run in the next
            raise
        self._early, self._in_step = None, True
        # from here until the final optimiser piece has been issued the bucket holds partial
        # sums: an exception in between (launch error, collective failure, KeyboardInterrupt)
        # must not leave it marked clean (fast_backward bypasses autograd's accumulate hook)
        self._grads_clean = False
        try:
            with ops.fast_backward():          # kernels add straight into the flat bucket
                out['loss'].backward(self._one)    # (a persistent 1: no ones_like fill per step)
                # side-stream finalizers (loss / SNR, anchor gradients) join the main stream
                # BEFORE the gradient reduction reads the bucket
                ops.join_deferred()
                if self._buckets is not None:
                    grad_scale = self._buckets.finish()            # pieces launched during backward
                else:                                              # ONE RCCL all-reduce / step
                    grad_scale = dist.allreduce_grads_(
                        self._grad_store if dist.is_dist() else self._flat_grad)
        finally:
            self._in_step = False
            chain.__exit__(None, None, None)
            ops.join_deferred()        # (no-op after a clean backward; an exception path still joins)
        self.step_count += 1
        ranges, early_stream = None, None
        if self._early is not None:            # everything but the bottom layer is already stepped
            early, early_stream = self._early
            ranges = dist._complement(early, self._flat_grad.numel())
            self._early = None
        self.ozer.step(self.step_count, self.learn_rate, clip=hparams.GRAD_CLIP_THRES,
                       grad_scale=grad_scale, zero_grad=not self.keep_grads, ranges=ranges)
        ops.repack_weights(self.device)    # operand-layout copies of the weights the GEMMs read
        if early_stream is not None:
            # joined AFTER the last piece (disjoint ranges): by now the early piece has long
            # finished, and a wait for an already signalled event costs nothing (waiting first
            # put a 20 us bubble in front of the last kernel of the step)
            torch.cuda.current_stream(self.device).wait_stream(early_stream)
        self._grads_clean = not self.keep_grads
        ops.step_done(self.device)
        if self._auto:
            self._auto_tick()
        return dict(loss=out['loss'].detach(), SNR=out['SNR'], LR=self.learn_rate)

    def collectives_per_step(self):
        '''data-path collectives a train step issues under data parallelism (0 without)'''
        if not dist.is_dist():
            return 0
        return {'0': 1, 'tail': 2}.get(self.grad_schedule,
                                       1 + hparams.NUM_LSTM_LAYERS + 1)   # '1': per-layer buckets

    def valid_step(self, s_src_signals):
        '''`g_sess.run(valid_fetches)` (main.py:499-500)'''
        ops.poll_status(self.device)
        with torch.no_grad():
            out = self.forward(s_src_signals, with_valid=True, with_train=False)
        ops.step_done(self.device, collective_consistent=not dist.is_dist())
        return dict(loss=out['valid_loss'], SNR=out['valid_SNR'])

    def infer(self, s_mixed_signals):
        '''`g_sess.run(infer_fetches, {s_mixed_signals: ...})` (main.py:384-385,
        685-690): complex mixture [B,T,F] -> separated complex [B,C,T,F] using
        the inference estimator and the mixture phase (main.py:333-335).'''
        B, E = hparams.BATCH_SIZE, hparams.EMBED_SIZE
        ops.poll_status(self.device)
        with torch.no_grad():
            fe = ops.frontend(s_mixed_signals[:, None].contiguous())
            s_embed = self.encoder(fe['mix_log'])
            s_attr = self.valid_estimator(s_embed, s_mix_pwr=fe['mix_pwr'])
            s_sep = self.separator(fe['mix_pwr'], s_attr, s_embed.reshape(B, -1, E))
            res = ops.reattach_phase(s_sep, fe['phasor'])
        ops.step_done(self.device, collective_consistent=not dist.is_dist())
        return res

    # ------------------------------------------------------- misc (main.py)
    def set_learn_rate(self, lr):
        self.learn_rate = float(lr)

    def get_learn_rate(self):
        return self.learn_rate

    def check_status(self):
        '''blocking check of the persistent kernels' hand-off status (end of an epoch / a
        sweep, before parameters or separated signals are written); raises DanetHipError
        after a timeout -- on every rank together under data parallelism'''
        ops.check_status(self.device)

    def status_words(self):
        '''the 4 spare floats of the gradient store (word 0 = hand-off status under data parallelism)'''
        return self._grad_store[:4]

    def zero_grad(self):
        self._flat_grad.zero_()
        self._grads_clean = True

    def reset_state(self):
        '''RNN states are never carried (main.py:538-540 re-zeros zeros)'''
        return

    def save_params(self, filename, step=None):
        '''trainable variables only, like tf.train.Saver(var_list=trainables)
        (main.py:357,399); stored as .npz keyed by the reference's TF variable
        names (a TF1 checkpoint cannot be written without TF).'''
        save_dir = os.path.dirname(os.path.abspath(filename))
        if not os.path.exists(save_dir):
            os.makedirs(save_dir)
        if step is not None:
            filename = '%s-%d' % (filename, step)
        np.savez(filename, **{k: v.detach().cpu().numpy() for k, v in self.vars.items()})

    def load_params(self, filename):
        if not filename.endswith('.npz'):
            filename = filename + '.npz'
        data = np.load(filename)
        with torch.no_grad():
            for k, v in self.vars.items():
                v.copy_(torch.as_tensor(data[k]).to(self.device))
        return True

    def weights_written(self):
        '''REQUIRED after any write to the parameters that torch's version counter does not see
        (`param.data` arithmetic, c10d collectives, an external optimizer or kernel writing
        through raw pointers): marks the operand-layout copies of the weights (ops.packed_weight,
        read by the projection / dYc / dX / gx products) stale, so the next product re-packs them.
        In-place torch ops on the variables themselves and this package's own optimizer step need
        no call.'''
        ops.weights_written(self._flat)

    def load_param_dict(self, di):
        '''set variables from {tf_variable_name: ndarray} (tests / oracle parity)'''
        with torch.no_grad():
            for k, a in di.items():
                self.vars[k].copy_(torch.as_tensor(np.asarray(a, dtype=np.float32)).to(self.device))

    def param_dict(self):
        return {k: v.detach().cpu().numpy() for k, v in self.vars.items()}

    def grad_dict(self):
        return {k: v.grad.detach().cpu().numpy() for k, v in self.vars.items()}
