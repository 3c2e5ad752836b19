#!/usr/bin/env python
'''Per-step wall time of the cfg-2 train step, each step synchronised (GPU box): looks for one-off
stalls (first status poll, allocator growth, ...) that a 24-step average would smear out.'''
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import __graft_entry__ as g
g.load_package()
from danet_amd.model import Model

class A: batch=32; layers=3; hdim=300; frames=128
hp = bench.setup_hparams(A, bench.CONFIGS['cfg2'])
dev = torch.device('cuda', 0)
batches = bench.make_batches(hp, 0, 4, dev)
model = Model('h', device=dev, seed=1337).build()
ts = []
for i in range(60):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    model.train_step(batches[i % 4])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    ts.append((1e3 * (t1 - t0), 1e3 * (t2 - t0)))
print('step: enqueue ms / complete ms')
for i, (a, b) in enumerate(ts):
    flag = '  <--' if (i > 2 and b > 4.5) else ''
    print('%3d  %7.3f  %7.3f%s' % (i, a, b, flag))

# unsynchronised blocks of 24 steps, like bench.py's timed region
print('blocks of 24 unsynchronised steps: ms per step')
for blk in range(12):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tq = []
    for i in range(24):
        a = time.perf_counter()
        model.train_step(batches[i % 4])
        tq.append(1e3 * (time.perf_counter() - a))
    torch.cuda.synchronize()
    dt = 1e3 * (time.perf_counter() - t0) / 24
    print('block %2d: %.3f ms/step   slowest enqueue %.2f ms (step %d)%s' % (
        blk, dt, max(tq), tq.index(max(tq)), '   <--' if dt > 3.7 else ''))
print('allocator: reserved %.1f MB, num_alloc_retries %d' % (
    torch.cuda.memory_reserved() / 1e6, torch.cuda.memory_stats().get('num_alloc_retries', 0)))
