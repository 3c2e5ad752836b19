#!/usr/bin/env python
'''Does it matter whether the RCCL group is created before or after the first HIP work
of the process?  usage: dist_order_probe.py first|after|warm  (GPU box)'''
import os
import sys
import time

import torch
import torch.distributed as tdist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

mode = sys.argv[1]
dev = torch.device('cuda', 0)
torch.cuda.set_device(0)
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29535')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')


def init():
    tdist.init_process_group('nccl', device_id=dev)


if mode == 'first':
    init()
if mode == 'warm':            # touch the device and create the side streams, then the group
    torch.zeros(1, device=dev)
    s = [torch.cuda.Stream(device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    init()
g.load_package()
import bench  # noqa: E402
from danet_amd.model import Model  # noqa: E402


class A:
    batch, frames, layers, hdim = 32, 128, 3, 300


hp = bench.setup_hparams(A)
batches = bench.make_batches(hp, 0, 2, dev)
model = Model('p', device=dev, seed=1).build()
for i in range(5):
    model.train_step(batches[i % 2])
torch.cuda.synchronize()
if mode == 'after':
    init()
for rep in range(2):
    t0 = time.perf_counter()
    for i in range(30):
        model.train_step(batches[i % 2])
    torch.cuda.synchronize()
    print('%s: %.3f ms/step' % (mode, 1e3 * (time.perf_counter() - t0) / 30), flush=True)
tdist.destroy_process_group()
