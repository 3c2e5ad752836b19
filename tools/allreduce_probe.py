#!/usr/bin/env python
'''Cost of the gradient all-reduce as issued by Model.train_step, with a 1-rank RCCL group
(GPU box): isolated latency for several message sizes.'''
import os
import torch
import torch.distributed as dist

os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('RANK', '0'); os.environ.setdefault('WORLD_SIZE', '1')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
for n in (1 << 10, 1 << 20, 6904920, 4 * 6904920):
    t = torch.randn(n, device='cuda')
    for _ in range(5):
        dist.all_reduce(t)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        dist.all_reduce(t)
    e1.record()
    torch.cuda.synchronize()
    print('all_reduce %10d floats: %.1f us' % (n, e0.elapsed_time(e1) * 1e3 / 20), flush=True)
dist.destroy_process_group()
