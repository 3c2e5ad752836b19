'''
Host -> HBM feed of the train / valid loops: what `feed_dict={s_src_signals: ...}` does in the
reference (main.py:417-431, :497-500), one batch AHEAD of the step that consumes it.

The reference builds every batch on the host (dataset iterator -> reshape -> random crop to
MAX_TRAIN_LEN, main.py:417-426) and hands it to `g_sess.run`, which copies it to the device
synchronously.  Here, once step i has been enqueued, the loop does for batch i+2 -- while the
device is still running the steps it has queued (the host enqueues a step in about a third of the
time the device needs for it; batch i+1 went through the same before step i was enqueued, so its
upload sits IN FRONT of step i's work on the upload stream):

    next(dataset iterator)
    draw the crop offset  randint(0, T-L-1)     (same `random` stream, same order as main.py:424-425)
    cast / crop / reshape into a PINNED staging slot   (one pass over the data, 0.3 ms at cfg 2;
                                                        only the cropped frames ever cross PCIe)
    hipMemcpyAsync into a fixed device slot on the upload stream + event

and the next step only makes the compute stream wait for that event.  No arithmetic happens here:
the cast real -> complex64 (toy data, main.py:418-421) is a numpy copy into the staging slot.

Measured and NOT kept (profiles/r04_feed_probe.txt, profiles/EXPERIMENTS.md): a feeder thread that
stages and / or uploads (the main thread's enqueue slows from 1.2 to 2.9 ms per step on the GIL and
the runtime's one-off 10-70 ms stall of a recurrent kernel comes back); an upload stream of its own
(HIP maps streams onto a handful of hardware queues: depending on creation order the third stream
shares a queue with the main or the side stream and the step is 1 ms slower -- the uploads ride on
the existing side stream, which is idle during the forward pass); device buffers from the caching
allocator (`record_stream`: re-allocation and event polling every step).
'''
from random import randint

import numpy as np
import torch

from .hparams import hparams


def to_batch_host(data_pt, crop_len=None):
    '''dataset batch [B*C, T, F] (real or complex) -> numpy view [B, C, T', F] + the crop applied
    (main.py:417-426).  Draws from python's `random` exactly when the reference does.'''
    a = np.asarray(data_pt[0])
    a = a.reshape(hparams.BATCH_SIZE, hparams.MAX_N_SIGNAL, -1, hparams.FEATURE_SIZE)
    if crop_len is not None and a.shape[2] > crop_len:
        beg = randint(0, a.shape[2] - crop_len - 1)                       # main.py:424-425
        a = a[:, :, beg:beg + crop_len]
    return a


class _Slot(object):
    '''one pinned staging buffer + one device buffer (both grow-only), the event of the last
    upload through them, and the event behind the last step that read the device buffer.  The
    two events are created once and re-recorded (no event object is created or destroyed per
    step: every runtime allocation inside the loop is a chance for the one-off stall of DESIGN 5)'''
    __slots__ = ('buf', 'dev', 'event', 'consumed', 'uploaded', 'read')

    def __init__(self):
        self.buf, self.dev, self.event, self.consumed = None, None, None, None
        self.uploaded = self.read = False      # has `event` / `consumed` been recorded yet

    def stage(self, a, pin=True):
        n = int(np.prod(a.shape))
        if self.uploaded:
            self.event.synchronize()           # the previous H2D copy out of this slot is done
        if self.buf is None or self.buf.numel() < n:
            self.buf = torch.empty(max(n, 1), dtype=torch.complex64)
            if pin:
                self.buf = self.buf.pin_memory()
        t = self.buf[:n].view(*a.shape)
        np.copyto(t.numpy(), a, casting='same_kind')      # cast + crop + gather in one pass
        return t


# staging slots are kept per device across feeds (a feed per epoch must not re-pin 25 MB of host
# memory every time); a second feed on the same device while one is active gets its own
_slot_cache = {}


def _take_slots(key, depth):
    ent = _slot_cache.get(key)
    if ent is None or ent['busy'] or len(ent['slots']) != depth:
        ent = dict(busy=False, slots=[_Slot() for _ in range(depth)])
        if key not in _slot_cache or not _slot_cache[key]['busy']:
            _slot_cache[key] = ent
    ent['busy'] = True
    return ent


class BatchFeed(object):
    '''for spectra in BatchFeed(dataset.epoch(...), device, crop_len): model.train_step(spectra)

    Yields complex64 [B, C, T', F] tensors on `device`, already ordered behind their upload on
    the stream that is current in the consumer.  A yielded tensor is a view of one of `depth`
    fixed device buffers: it stays valid until the consumer asks for the next batch but `depth - 2`
    (copy it to keep it longer).

    mode (default: env DANET_FEED_MODE, else 'ahead' on a GPU and 'sync' on a CPU device):
      'sync'   the reference's literal form: convert, blocking upload from pageable memory
      'ahead'  through pinned staging slots, uploads on ops.copy_stream(), issued one step early
               (batch i+1 is on its way before step i is enqueued)'''

    def __init__(self, source, device, crop_len=None, depth=3, mode=None):
        import os
        self.source = source
        self.device = torch.device(device)
        if self.device.type == 'cuda' and self.device.index is None:
            self.device = torch.device('cuda', torch.cuda.current_device())
        self.cuda = self.device.type == 'cuda'
        self.crop_len = crop_len
        if mode is None:
            mode = os.environ.get('DANET_FEED_MODE', 'ahead') if self.cuda else 'sync'
        assert mode in ('sync', 'ahead'), mode
        self.mode = mode
        self.depth = max(3, depth)
        self.n = 0
        self._out = None
        self._k = 0
        if mode == 'ahead':
            self._ent = _take_slots(str(self.device), self.depth)
            self.slots = self._ent['slots']
            self.copy_stream = None
            if self.cuda:
                from . import ops
                self.copy_stream = ops.copy_stream(self.device)

    def _stage(self, data_pt):
        '''host batch -> (pinned staging tensor, its slot); draws the crop offset'''
        a = to_batch_host(data_pt, self.crop_len)
        slot = self.slots[self._k % self.depth]
        self._k += 1
        return slot.stage(a, pin=self.cuda), slot

    def _upload(self, t, slot):
        '''issue the H2D copy of a staged batch on the upload stream -> (device tensor, slot)'''
        if not self.cuda:                       # (host-logic tests: the "upload" is a copy)
            return t.clone(), slot
        n = t.numel()
        if slot.dev is None or slot.dev.numel() < n:
            slot.dev = torch.empty(n, dtype=torch.complex64, device=self.device)
            torch.cuda.current_stream(self.device).synchronize()     # (growth only)
        d = slot.dev[:n].view(t.shape)
        if slot.event is None:
            slot.event, slot.consumed = torch.cuda.Event(), torch.cuda.Event()
        with torch.cuda.stream(self.copy_stream):
            if slot.read:                       # the step that read this buffer last has finished
                self.copy_stream.wait_event(slot.consumed)
            d.copy_(t, non_blocking=True)
            slot.event.record(self.copy_stream)
        slot.uploaded = True
        return d, slot

    def _hand_out(self, d, slot):
        if self.cuda:
            torch.cuda.current_stream(self.device).wait_event(slot.event)
        self._out = slot
        self.n += 1
        return d

    def _consumed(self):
        '''the consumer is back: everything it enqueued that reads the last batch is on its
        stream now -- mark the point after which that batch's device buffer may be rewritten'''
        slot, self._out = self._out, None
        if slot is not None and self.cuda:
            slot.consumed.record(torch.cuda.current_stream(self.device))
            slot.read = True

    def __iter__(self):
        if self.mode == 'sync':
            for data_pt in self.source:
                a = to_batch_host(data_pt, self.crop_len)
                self.n += 1
                yield torch.as_tensor(np.ascontiguousarray(a).astype(np.complex64)).to(self.device)
            return
        it = iter(self.source)

        def fetch():
            try:
                return self._upload(*self._stage(next(it)))
            except StopIteration:
                return None
            except Exception as e:     # raised when the batch's turn comes, not one batch early
                return e
        try:
            # the upload of batch i+1 is ISSUED before the consumer enqueues step i, so on the
            # device it runs during step i's forward pass (the upload stream is idle then) and not
            # behind step i's side-stream work -- with a 2.8 ms step an upload queued behind the
            # last weight-gradient group finished only just before the step did.  Batch i+2 is
            # staged (0.3 ms of host time) while step i runs; its upload waits on the device for
            # the step that last read its buffer (i-1).
            def live(x):
                return x is not None and not isinstance(x, Exception)
            cur = fetch()
            nxt = fetch() if live(cur) else None
            while cur is not None:
                if isinstance(cur, Exception):
                    raise cur
                yield self._hand_out(*cur)        # the consumer enqueues step i ...
                self._consumed()
                cur = nxt
                nxt = fetch() if live(cur) else None         # ... then batch i+2 is staged and issued
        finally:
            self._consumed()                      # (a consumer that left the loop early: its last
            self._ent['busy'] = False             # batch's buffer is protected all the same)


class StepReport(object):
    '''cli_report of main.py:413-436 without a host synchronisation per step: the fetched
    metrics are kept as they come (device scalars stay on the device) and summed on the host, in
    step order, when the epoch mean is read -- ONE device read per `flush_every` steps.  The
    result equals the reference's running `dst[k] += float(v)` sum bit for bit (same values, same
    order of double additions).'''

    def __init__(self, flush_every=1024):
        self.sums, self.pending, self.n = {}, {}, 0
        self.keys = []
        self.flush_every = flush_every

    def add(self, fetch):
        for k, v in fetch.items():
            if k not in self.pending:
                self.pending[k] = []
                self.keys.append(k)
            self.pending[k].append(v)
        self.n += 1
        if self.n % self.flush_every == 0:
            self.flush()

    def flush(self):
        for k, li in self.pending.items():
            tens = [v.detach().reshape(()) for v in li if torch.is_tensor(v)]
            vals = iter(torch.stack(tens).cpu().tolist()) if tens else iter(())
            s = self.sums.get(k, 0.)
            for v in li:
                s = s + (next(vals) if torch.is_tensor(v) else float(v))
            self.sums[k] = s
            li[:] = []

    def mean(self):
        '''OrderedDict-like list of (key, epoch mean) in first-seen key order'''
        from collections import OrderedDict
        self.flush()
        return OrderedDict((k, self.sums[k] * (1. / max(self.n, 1))) for k in self.keys)
