#!/usr/bin/env python
'''Is the train step GPU-bound or host(launch)-bound?  Times how long the host
needs to ENQUEUE K steps vs how long the GPU needs to EXECUTE them.'''
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import __graft_entry__ as g
g.load_package()
from danet_amd.model import Model

class A: batch=32; layers=3; hdim=300; frames=128
hp = bench.setup_hparams(A, bench.CONFIGS["cfg2"])
dev = torch.device('cuda', 0)
batches = bench.make_batches(hp, 0, 2, dev)
model = Model('h', device=dev).build()
for i in range(5):
    model.train_step(batches[i % 2])
torch.cuda.synchronize()
K = 20
t0 = time.perf_counter()
for i in range(K):
    model.train_step(batches[i % 2])
t_issue = time.perf_counter() - t0
torch.cuda.synchronize()
t_all = time.perf_counter() - t0
print('enqueue %.3f ms/step, complete %.3f ms/step' % (1e3 * t_issue / K, 1e3 * t_all / K))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for i in range(5):
    model.train_step(batches[i % 2])
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
