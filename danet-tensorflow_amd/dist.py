'''
Data-parallel plumbing (new: the reference is single-GPU, README.md:226,
main.py:584 "TODO manage device").

One process per GPU, `hparams.BATCH_SIZE` mixtures per process; mixtures are
independent through the whole forward and the loss is a mean over the batch
(app/ops.py:430), so the global-batch gradient is the mean of the per-rank
gradients.  The only collective on the data path is ONE all-reduce(SUM) per
step over the model's flat fp32 gradient bucket (RCCL over xGMI; `nccl` backend
on ROCm), the 1/world scale is folded into the optimiser kernel, and the value
clip (main.py:359-362) is applied AFTER the reduction so it keeps its
global-batch meaning.  The functions take plain tensors, so the same code runs
under `gloo` on CPU tensors in the tests.
'''
import os

import torch
import torch.distributed as dist


def is_dist():
    return dist.is_available() and dist.is_initialized()


def world_size():
    return dist.get_world_size() if is_dist() else 1


def rank():
    return dist.get_rank() if is_dist() else 0


def init_from_env(backend='nccl', device=None):
    '''torch.distributed.run / torchrun environment (RANK, WORLD_SIZE,
    MASTER_ADDR, MASTER_PORT); no-op for a single process.'''
    if int(os.environ.get('WORLD_SIZE', '1')) <= 1 or is_dist():
        return
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    kw = {}
    if backend == 'nccl' and device is not None:
        kw['device_id'] = device
    dist.init_process_group(backend, **kw)


def broadcast_params_(flat, src=0):
    '''identical initial parameters on every rank'''
    if is_dist():
        dist.broadcast(flat, src=src)
    return flat


def allreduce_grads_(flat_grad):
    '''SUM the flat gradient bucket over ranks (in place); returns the factor
    the caller must multiply by to get the global-batch mean gradient.'''
    w = world_size()
    if is_dist():          # also at w == 1 (keeps the RCCL path exercised)
        dist.all_reduce(flat_grad, op=dist.ReduceOp.SUM)
    return 1.0 / w


def allreduce_max_scalar(x, device):
    if world_size() == 1:
        return float(x)
    t = torch.tensor([float(x)], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_seed(base_seed):
    '''a different synthetic shard per rank (seed = 1337 + rank)'''
    return base_seed + rank()
