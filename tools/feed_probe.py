'''GPU probe: ms per cfg-2 train step through cli.train_epoch for every feed mode (feed.BatchFeed)
next to the HBM-resident loop.  python tools/feed_probe.py [steps]'''
import io, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g; g.load_package()
import bench
from danet_amd import cli, feed, ops
from danet_amd.model import Model
import types

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
args = types.SimpleNamespace(batch=32, frames=128, layers=3, hdim=300)
hp = bench.setup_hparams(args, bench.CONFIGS['cfg2'])
dev = torch.device('cuda', 0)
model = Model('probe', device=dev, seed=1337).build()
res = bench.make_batches(hp, 0, 4, dev)
host = bench.make_host_batches(hp, 0, 4, dev)
host128 = [np.ascontiguousarray(h[:, :128]) for h in host]
for i in range(12):
    model.train_step(res[i % 4])
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    model.train_step(res[i % 4])
torch.cuda.synchronize()
print('resident            %.3f ms/step' % (1e3 * (time.perf_counter() - t0) / steps), flush=True)


def run(mode, batches, sw=None, flush=1024):
    if sw is not None:
        sys.setswitchinterval(sw)
    def epoch(n):
        for i in range(n):
            yield (batches[i % 4],)
    def loop(n):
        src = feed.BatchFeed(epoch(n), dev, hp.MAX_TRAIN_LEN, mode=mode)
        rep = feed.StepReport(flush_every=flush)
        for x in src:
            rep.add(model.train_step(x))
        return rep.mean()
    loop(8)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loop(steps)
    torch.cuda.synchronize()
    dt = 1e3 * (time.perf_counter() - t0) / steps
    sys.setswitchinterval(0.005)
    return dt

# which stream the uploads ride on: HIP maps streams onto few hardware queues
for cs in ('side', 'own', 'own', 'own', 'side', 'side'):
    ops._copy.clear()
    ops.COPY_STREAM = cs
    print('copy stream = %-4s: ahead %.3f  ahead %.3f  sync %.3f ms/step'
          % (cs, run('ahead', host), run('ahead', host128), run('sync', host)), flush=True)
print('sync, metrics deferred %.3f' % run('sync', host, flush=1024), flush=True)
print('ahead, metrics per step %.3f' % run('ahead', host, flush=1), flush=True)
# host-side cost of staging alone
sl = feed._Slot()
a = feed.to_batch_host((host[0],), 128)
t0 = time.perf_counter()
for i in range(20):
    sl.stage(feed.to_batch_host((host[i % 4],), 128), pin=True)
print('stage (crop+copy into pinned) %.3f ms' % (1e3 * (time.perf_counter() - t0) / 20))
t0 = time.perf_counter()
for i in range(20):
    np.ascontiguousarray(a).astype(np.complex64)
print('ascontiguous+astype %.3f ms' % (1e3 * (time.perf_counter() - t0) / 20))
assert ops.lstm_status_ok()
