#!/usr/bin/env python
'''Feasibility probe: does running two half-batches concurrently on two HIP
streams beat one full batch?  (Two independent B=16 models vs one B=32 model.)'''
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import __graft_entry__ as g
g.load_package()
from danet_amd.model import Model
from danet_amd import ops


def run(nmodels, batch, steps=30, skew=False):
    class A: layers = 3; hdim = 300; frames = 128
    A.batch = batch
    hp = bench.setup_hparams(A)
    dev = torch.device('cuda', 0)
    models = [Model('m%d' % i, device=dev, seed=1 + i).build() for i in range(nmodels)]
    data = [bench.make_batches(hp, i, 2, dev) for i in range(nmodels)]
    streams = [torch.cuda.Stream() for _ in range(nmodels)]
    def step(k):
        for i, m in enumerate(models):
            with torch.cuda.stream(streams[i]):
                m.train_step(data[i][k % 2])
    for k in range(5):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        step(k)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    assert ops.lstm_status_ok()
    mixs = nmodels * batch * 128 * 64 / 8000.0
    print('%d model(s) x B=%d: %.3f ms per (all-models) step -> %.0f mixture-seconds/s' % (
        nmodels, batch, 1e3 * dt, mixs / dt))


if __name__ == '__main__':
    run(1, 32)
    run(2, 16)
    run(1, 16)
    run(4, 8)
