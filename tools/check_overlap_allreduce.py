#!/usr/bin/env python
'''1-rank RCCL check of the opt-in overlapped gradient all-reduce: after 3 train
steps the parameters must be bit-identical to the single-all-reduce path.
Run with DANET_FORCE_DIST=1 (see bench.py) on a GPU box.'''
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
class A: batch = 32; layers = 3; hdim = 300; frames = 128
hp = bench.setup_hparams(A)
batches = bench.make_batches(hp, 0, 2, dev)
res = {}
for mode in ('0', '1'):
    os.environ['DANET_OVERLAP_ALLREDUCE'] = mode
    del ops.GRAD_READY_HOOKS[:]
    m = Model('o' + mode, device=dev, seed=5).build()
    assert (m._buckets is not None) == (mode == '1')
    for k in range(3):
        m.train_step(batches[k % 2])
    torch.cuda.synchronize()
    if mode == '1':
        print('buckets reduced per step:', 'hooks registered =', len(ops.GRAD_READY_HOOKS))
    res[mode] = m.param_dict()
assert ops.lstm_status_ok()
worst = max(np.abs(res['0'][k] - res['1'][k]).max() for k in res['0'])
print('max |param diff| overlap vs single all-reduce after 3 steps:', worst)
assert worst == 0.0
torch.distributed.destroy_process_group()
print('OK')
