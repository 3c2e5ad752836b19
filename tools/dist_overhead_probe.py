#!/usr/bin/env python
'''Why is a train step slower with a (1-rank) RCCL group?  (GPU box)'''
import os
import sys
import time

import torch
import torch.distributed as tdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.load_package()
import bench  # noqa: E402
from danet_amd import dist as ddist  # noqa: E402
from danet_amd.model import Model  # noqa: E402


class A:
    batch, frames, layers, hdim = 32, 128, 3, 300


def run(model, batches, K=30):
    for i in range(5):
        model.train_step(batches[i % 2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(K):
        model.train_step(batches[i % 2])
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / K


hp = bench.setup_hparams(A)
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
batches = bench.make_batches(hp, 0, 2, dev)
model = Model('p', device=dev, seed=1).build()
print('no process group            : %.3f ms/step' % run(model, batches), flush=True)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29534')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
tdist.init_process_group('nccl', device_id=dev)
orig = ddist.allreduce_grads_
ddist.allreduce_grads_ = lambda g_: 1.0
print('group initialised, no call  : %.3f ms/step' % run(model, batches), flush=True)
ddist.allreduce_grads_ = orig
print('one all-reduce per step     : %.3f ms/step' % run(model, batches), flush=True)
tdist.all_reduce(torch.zeros(4, device=dev)); torch.cuda.synchronize()
print('again                       : %.3f ms/step' % run(model, batches), flush=True)
model2 = Model('q', device=dev, seed=1).build()      # built AFTER the group exists (as bench.py does)
print('model built under the group : %.3f ms/step' % run(model2, batches), flush=True)
ddist.allreduce_grads_ = lambda g_: 1.0
print('   ... without the call     : %.3f ms/step' % run(model2, batches), flush=True)
ddist.allreduce_grads_ = orig
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    tdist.barrier()
    torch.cuda.synchronize()
    print('dist.barrier() + synchronize: %.3f ms' % (1e3 * (time.perf_counter() - t0)), flush=True)
tdist.destroy_process_group()
