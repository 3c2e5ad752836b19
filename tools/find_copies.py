#!/usr/bin/env python
'''List the aten copy/fill ops of one train step with their shapes (GPU box).'''
import os
import sys
import collections

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

g.load_package()
import bench  # noqa: E402
from danet_amd.model import Model  # noqa: E402


class A:
    batch, frames, layers, hdim = 32, 128, 3, 300


hp = bench.setup_hparams(A)
dev = torch.device('cuda')
batches = bench.make_batches(hp, 0, 2, dev)
model = Model('probe', device=dev, seed=1).build()
for i in range(3):
    model.train_step(batches[i % 2])
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU], record_shapes=True, with_stack=False) as prof:
    model.train_step(batches[0])
    torch.cuda.synchronize()
c = collections.Counter()
for e in prof.events():
    if e.name in ('aten::copy_', 'aten::clone', 'aten::contiguous', 'aten::zero_', 'aten::fill_', 'aten::zeros',
                  'aten::add', 'aten::add_', 'aten::mul', 'aten::sum', 'aten::ones_like', 'aten::zeros_like'):
        c[(e.name, str(e.input_shapes)[:90])] += 1
for k, v in sorted(c.items()):
    print(v, k)
