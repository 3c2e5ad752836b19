#!/usr/bin/env python
'''Where inside a train step does the sporadic 17-33 ms stall sit?  Every library call of the first
steps after a synchronisation gets an event pair; prints calls / gaps longer than 3 ms.
Run repeatedly (fresh process each time) on a GPU box.'''
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import __graft_entry__ as g
g.load_package()
from danet_amd.model import Model
from danet_amd import _lib

class A: batch=32; layers=3; hdim=300; frames=128
hp = bench.setup_hparams(A, bench.CONFIGS['cfg2'])
dev = torch.device('cuda', 0)
batches = bench.make_batches(hp, 0, 4, dev)
model = Model('h', device=dev, seed=1337).build()
model.train_step(batches[0])
for i in range(5):
    model.train_step(batches[i % 4])
torch.cuda.synchronize()
marks = []
orig_enter, orig_exit = _lib.timed.__enter__, _lib.timed.__exit__
def enter(self):
    e = torch.cuda.Event(enable_timing=True); e.record(); self._e = e
    return self
def exit_(self, *exc):
    e = torch.cuda.Event(enable_timing=True); e.record()
    marks.append((self.label + (':' + self.tag if self.tag else ''), self._e, e, torch.cuda.current_stream().cuda_stream))
    return False
_lib.timed.__slots__ = ()
class T2(object):
    def __init__(self, label, tag=None): self.label, self.tag = label, tag
    __enter__ = enter
    __exit__ = exit_
_lib.timed = T2
import danet_amd.ops as ops
step_ev = [torch.cuda.Event(enable_timing=True)]; step_ev[0].record()
for i in range(14):
    model.train_step(batches[i % 4])
    e = torch.cuda.Event(enable_timing=True); e.record(); step_ev.append(e)
torch.cuda.synchronize()
st = [step_ev[i].elapsed_time(step_ev[i + 1]) for i in range(14)]
print('steps ms:', ' '.join('%.2f' % x for x in st))
if max(st) > 6:
    base = step_ev[0]
    rows = [(base.elapsed_time(a), base.elapsed_time(b), lab, s) for lab, a, b, s in marks]
    rows.sort()
    prev_end = {}
    for a, b, lab, s in rows:
        if b - a > 3:
            print('LONG call %-28s start %.2f ms dur %.2f ms stream %x' % (lab, a, b - a, s))
        pe = prev_end.get(s)
        if pe is not None and a - pe > 3:
            print('GAP before %-28s at %.2f ms: %.2f ms (same stream %x)' % (lab, a, a - pe, s))
        prev_end[s] = b
