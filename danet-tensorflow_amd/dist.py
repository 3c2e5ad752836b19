'''
Data-parallel plumbing (new: the reference is single-GPU, README.md:226,
main.py:584 "TODO manage device").

One process per GPU, `hparams.BATCH_SIZE` mixtures per process; mixtures are
independent through the whole forward and the loss is a mean over the batch
(app/ops.py:430), so the global-batch gradient is the mean of the per-rank
gradients.  The only collective on the data path is ONE all-reduce(SUM) per
step over the model's flat fp32 gradient bucket (RCCL over xGMI; `nccl` backend
on ROCm), the 1/world scale is folded into the optimiser kernel, and the value
clip (main.py:359-362) is applied AFTER the reduction so it keeps its
global-batch meaning.  The functions take plain tensors, so the same code runs
under `gloo` on CPU tensors in the tests.
'''
import os

import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if is_dist() else 1


def rank():
    return dist.get_rank() if is_dist() else 0


def init_from_env(backend='nccl', device=None):
    '''torch.distributed.run / torchrun environment (RANK, WORLD_SIZE,
    MASTER_ADDR, MASTER_PORT); no-op for a single process.'''
    if int(os.environ.get('WORLD_SIZE', '1')) <= 1 or is_dist():
        return
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    kw = {}
    if backend == 'nccl' and device is not None:
        kw['device_id'] = device
        from . import ops
        ops.prepare_streams(device)       # side streams BEFORE the RCCL communicator
    dist.init_process_group(backend, **kw)


def broadcast_params_(flat, src=0):
    '''identical parameters on every rank.  A c10d collective writes `flat` without touching
    torch's version counter, so the operand-layout copies of the weights (ops.packed_weight) are
    marked stale HERE -- otherwise a receiving rank would keep multiplying with its pre-broadcast
    packs (e.g. the NaN weights a restore was meant to replace).'''
    if is_dist():
        dist.broadcast(flat, src=src)
        from . import ops
        ops.weights_written(flat)
    return flat


def allreduce_grads_(flat_grad):
    '''SUM the flat gradient bucket over ranks (in place); returns the factor
    the caller must multiply by to get the global-batch mean gradient.'''
    w = world_size()
    if is_dist():          # also at w == 1 (keeps the RCCL path exercised)
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return 1.0 / w


def allreduce_mean_scalars(values, device):
    '''mean over ranks of a list of python floats (epoch metrics: every rank must take the
    same learning-rate / NaN-restore decisions, main.py:439-476); NaN on any rank -> NaN'''
    if world_size() == 1:
        return [float(v) for v in values]
    t = torch.tensor([float(v) for v in values], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return [float(v) / world_size() for v in t.tolist()]


def allreduce_max_scalar(x, device):
    if world_size() == 1:
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_seed(base_seed):
    '''a different synthetic shard per rank (seed = 1337 + rank)'''
    return base_seed + rank()


def _complement(covered, n):
    '''maximal [lo, hi) pieces of [0, n) not in the sorted-able list `covered`'''
    out, pos = [], 0
    for lo, hi in sorted(covered) + [(n, n)]:
        if lo > pos:
            out.append((pos, lo))
        pos = max(pos, hi)
    return out


# ---- which reduction schedule?  (Model(grad_schedule='auto'), the default) ---------------------------
# '0' (ONE all-reduce after backward, the north_star form) exposes the whole collective.  'tail' launches
# everything outside the bottom encoder layer's share of the bytes (85 % at cfg 2) on the side stream once
# that layer's BPTT kernel has finished, so that it runs under that layer's weight-gradient GEMMs (ordinary
# tile kernels; it deliberately never runs beside a persistent recurrent kernel), and reduces the rest -- with
# the status words -- after backward.  What it can hide is bounded by that cover, ~3 % of the step (75 us at
# cfg 2, 0.3 ms at cfg 4 as written); what it costs is a second collective launch (~20-30 us) and a
# cross-stream wait on the main stream (~10 us).  So it pays once the collective is about as long as the cover:
# ONE threshold on  (stand-alone all-reduce time) / (step time),  both MEASURED by the model on its own bucket
# and its own steps and MAX-reduced over the ranks, so that every rank takes the same decision.  Default 5 %;
# above it 'tail' is worth cover - 0.04 ms = 1-3 points of scaling efficiency, below it the
# plain north_star form stays.  DANET_ALLREDUCE_TAIL_RATIO overrides the threshold, DANET_OVERLAP_ALLREDUCE /
# Model(grad_schedule=...) pins the schedule.
TAIL_RATIO = float(os.environ.get('DANET_ALLREDUCE_TAIL_RATIO', '0.05'))


def choose_schedule(allreduce_ms, step_ms, ratio=None):
    '''-> '0' | 'tail' from the measured stand-alone all-reduce time of the gradient bucket and the measured
    train-step time (both already identical on every rank)'''
    r = TAIL_RATIO if ratio is None else float(ratio)
    if not (allreduce_ms >= 0.0 and step_ms > 0.0):        # (NaN / nonsense: stay with the plain form)
        return '0'
    return 'tail' if allreduce_ms > r * step_ms else '0'


def measure_allreduce_ms(numel, device, reps=3):
    '''stand-alone all-reduce time of a scratch fp32 bucket of `numel` elements (a collective: every rank
    calls it at the same point), MAX over ranks; 0.0 without a process group'''
    if not is_dist():
        return 0.0
    t = torch.zeros(int(numel), dtype=torch.float32, device=device)
    dist.all_reduce(t)                                      # (communicator / buffers warm)
    if t.is_cuda:
        torch.cuda.synchronize(device)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            dist.all_reduce(t)
        e1.record()
        torch.cuda.synchronize(device)
        ms = e0.elapsed_time(e1) / reps
    else:
        import time
        t0 = time.perf_counter()
        for _ in range(reps):
            dist.all_reduce(t)
        ms = (time.perf_counter() - t0) * 1e3 / reps
    return allreduce_max_scalar(ms, device)


class TailOverlap(object):
    '''The 'tail' gradient reduction schedule (opt-in, or picked by Model(grad_schedule='auto') when the
    measured all-reduce is long against the measured step: `choose_schedule`).

    When the BOTTOM encoder layer's BPTT kernel has been issued, every gradient except
    that layer's own is final (backward runs top-down and nothing with parameters sits
    below the encoder).  ops fires ('rest', bottom_params) on a side stream that has waited
    for the main stream and the upper layers' weight-gradient chains; the hook launches
    ONE asynchronous all-reduce over everything outside the bottom layer's range
    (23.5 of 27.6 MB at cfg 2), which then runs under the bottom layer's weight-gradient
    GEMMs -- ordinary tile kernels, so unlike the per-layer schedule (GradBuckets) the
    collective never shares the GPU with a persistent recurrent kernel whose workgroups
    must stay co-resident.  `finish()` reduces the bottom layer's range after backward and
    waits for both.  Every rank issues the same collectives in the same order; without a
    ('rest',) event (another encoder type) finish() reduces the whole bucket at once.'''

    def __init__(self, flat_grad, offsets, last=()):
        '''offsets: {param data_ptr: (start, end)} element ranges in flat_grad; last: ranges that must
        ride in the LAST collective of the step (the hand-off status words: see Model._flatten)'''
        self.flat = flat_grad
        self.offsets = offsets
        self.last = list(last)
        self.works = []
        self.covered = []
        self.fired = False
        self.launched = 0        # pieces launched from hooks so far (diagnostics / tests)

    def hook(self, tag, params):
        if tag[0] != 'rest' or self.fired or not is_dist():
            return
        rng = [self.offsets[p.data_ptr()] for p in params if p.data_ptr() in self.offsets]
        if len(rng) != len(params):
            return                      # not this model's encoder
        self.fired = True
        n = self.flat.numel()
        for lo, hi in _complement(rng + self.last, n):
            self.works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM,
                                              async_op=True))
            self.covered.append((lo, hi))
            self.launched += 1

    def wait_launched(self):
        '''the current stream waits for the collectives launched so far (their ranges are then
        final on it: Model steps the optimizer over them early)'''
        for wk in self.works:
            wk.wait()

    def finish(self):
        '''returns the 1/world factor like allreduce_grads_'''
        w = world_size()
        if is_dist():
            for lo, hi in _complement(self.covered, self.flat.numel()):
                self.works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM,
                                                  async_op=True))
            for wk in self.works:
                wk.wait()
        self.works, self.covered, self.fired = [], [], False
        return 1.0 / w


class GradBuckets(object):
    '''Overlapped gradient reduction (opt-in: DANET_OVERLAP_ALLREDUCE=1).

    The flat gradient bucket is reduced in contiguous pieces as backward produces
    them -- output projection first, then the LSTM layers from the top down --
    each as one asynchronous all-reduce launched behind that piece's
    weight-gradient GEMMs (ops.GRAD_READY_HOOKS), so the collectives run under the
    remaining BPTT kernels.  `finish()` reduces whatever was not covered (the tiny
    estimator variables) and makes the current stream wait for every piece.
    Every rank issues the same collectives in the same order.'''

    def __init__(self, flat_grad, offsets, last=()):
        '''offsets: {param data_ptr: (start, end)} element ranges in flat_grad; last: ranges for
        finish() (never covered by a hook's piece: the status words)'''
        self.flat = flat_grad
        self.offsets = offsets
        self.last = list(last)
        self.works = []
        self.covered = []
        self.launched = 0

    def hook(self, tag, params):
        if tag[0] == 'rest' or not is_dist():
            return
        rng = [self.offsets[p.data_ptr()] for p in params if p.data_ptr() in self.offsets]
        if len(rng) != len(params):
            return                      # not ours (e.g. a stand-alone layer call)
        lo, hi = min(r[0] for r in rng), max(r[1] for r in rng)
        if sum(r[1] - r[0] for r in rng) != hi - lo:
            return                      # not contiguous: leave it to finish()
        self.works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM, async_op=True))
        self.covered.append((lo, hi))
        self.launched += 1

    def finish(self):
        '''returns the 1/world factor like allreduce_grads_'''
        w = world_size()
        if is_dist():
            for lo, hi in _complement(self.covered, self.flat.numel()):
                self.works.append(dist.all_reduce(self.flat[lo:hi], op=dist.ReduceOp.SUM,
                                                  async_op=True))
            for wk in self.works:
                wk.wait()
        self.works, self.covered = [], []
        return 1.0 / w
