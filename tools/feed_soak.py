import io, os, sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
import __graft_entry__ as g; g.load_package()
import bench, types
from danet_amd import cli, ops
from danet_amd.model import Model
args = types.SimpleNamespace(batch=32, frames=128, layers=3, hdim=300)
hp = bench.setup_hparams(args, bench.CONFIGS['cfg2'])
dev = torch.device('cuda', 0)
model = Model('soak', device=dev, seed=1337).build()
host = bench.make_host_batches(hp, 0, 4, dev)
def epoch(n):
    for i in range(n):
        yield (host[i % 4],)
cli.train_epoch(model, epoch(20), io.StringIO())
torch.cuda.synchronize()
m0 = torch.cuda.memory_reserved()
for rep in range(3):
    t0 = time.perf_counter()
    rep_, n = cli.train_epoch(model, epoch(3000), io.StringIO())
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('epoch of %d steps: %.3f ms/step, loss %.4f, reserved %.1f MB (start %.1f)' % (
        n, 1e3 * dt / n, rep_['loss'], torch.cuda.memory_reserved() / 2**20, m0 / 2**20), flush=True)
model.check_status()
