#!/usr/bin/env python
'''1-rank RCCL check of the gradient-reduction schedules (dist.py), `check_overlap_allreduce.py [N [cfg2|cfg4h600]]`:
after N train steps (argv[1], default 3) of the configuration argv[2] (default cfg2; cfg4h600 = BASELINE configs[3]
as written, 142.5 MB bucket) the
parameters of 'tail' (opt-in: everything but the bottom layer reduced under the bottom
layer's weight-gradient GEMMs) and '1' (per-layer buckets) must be bit-identical to '0' (one
all-reduce after backward), and the expected number of asynchronous pieces must have been
launched from the gradient-ready hooks.  (A 1-rank all-reduce is an identity, so the STREAM
ORDER of the pieces is checked separately with a doubling stand-in collective:
tests/test_gpu_extensions.py::test_reduction_schedules_respect_stream_order.)'''
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
import numpy as np
import torch
import bench
import __graft_entry__ as g
g.load_package()
from danet_amd.model import Model
from danet_amd import ops

dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
ops.prepare_streams(dev)          # side streams before the RCCL communicator
torch.distributed.init_process_group('nccl', device_id=dev)
CFG = sys.argv[2] if len(sys.argv) > 2 else 'cfg2'
class A: pass
A.batch, A.layers, A.hdim, A.frames = (bench.CONFIGS[CFG][k] for k in ('batch', 'layers', 'hdim', 'frames'))
hp = bench.setup_hparams(A, bench.CONFIGS[CFG])
batches = bench.make_batches(hp, 0, 2, dev)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3
NPIECES = 1 + A.layers          # per-layer buckets: the output projection + one per layer
res = {}
for mode in ('0', 'tail', '1'):
    m = Model('o' + mode, device=dev, seed=5, grad_schedule=mode).build()
    m._early_adam = True          # the opt-in early optimizer piece rides behind the tail all-reduce
    assert (m._buckets is not None) == (mode != '0')
    assert m.collectives_per_step() == {'0': 1, 'tail': 2, '1': NPIECES + 1}[mode]
    for k in range(N):
        m.train_step(batches[k % 2])
    torch.cuda.synchronize()
    if m._buckets is not None:
        print('mode', mode, 'asynchronous pieces launched from hooks in %d steps:' % N, m._buckets.launched)
        assert m._buckets.launched == (N if mode == 'tail' else N * NPIECES), m._buckets.launched
    # the early optimizer piece runs behind the tail all-reduce only in the 'tail' schedule
    assert m.early_steps == (N if mode == 'tail' else 0), (mode, m.early_steps)
    res[mode] = m.param_dict()
    del m
assert ops.lstm_status_ok()
for mode in ('tail', '1'):
    worst = max(np.abs(res['0'][k] - res[mode][k]).max() for k in res['0'])
    print('max |param diff| %s vs single all-reduce after %d steps:' % (mode, N), worst)
    assert worst == 0.0
# 'auto' under the 1-rank group: the measured all-reduce (an identity: ~0.01 ms) is far below the threshold
m = Model('oauto', device=dev, seed=5).build() if 'DANET_OVERLAP_ALLREDUCE' not in os.environ else None
if m is not None:
    for k in range(max(N, Model.AUTO_DECIDE_AT + 1)):
        m.train_step(batches[k % 2])
    torch.cuda.synchronize()
    print('auto decision:', m.schedule_decision)
    assert m.schedule_decision is not None and m.schedule_decision['schedule'] == '0' == m.grad_schedule
    if N >= Model.AUTO_DECIDE_AT + 1:
        worst = max(np.abs(res['0'][k] - v).max() for k, v in m.param_dict().items())
        assert worst == 0.0, worst       # the decision (collectives on scratch buffers) leaves the run untouched
torch.distributed.destroy_process_group()
print('OK')
