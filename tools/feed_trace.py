'''GPU probe: per-step host and device times of the train loop through feed.BatchFeed (why is the
loop bimodal?).  python tools/feed_trace.py [mode] [steps] [repeats]'''
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import __graft_entry__ as g; g.load_package()
import bench, types
from danet_amd import feed, ops
from danet_amd.model import Model

mode = sys.argv[1] if len(sys.argv) > 1 else 'ahead'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 80
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
args = types.SimpleNamespace(batch=32, frames=128, layers=3, hdim=300)
hp = bench.setup_hparams(args, bench.CONFIGS['cfg2'])
dev = torch.device('cuda', 0)
model = Model('probe', device=dev, seed=1337).build()
host = bench.make_host_batches(hp, 0, 4, dev)
res = bench.make_batches(hp, 0, 4, dev)
for i in range(12):
    model.train_step(res[i % 4])
torch.cuda.synchronize()


def epoch(n):
    for i in range(n):
        yield (host[i % 4],)


import gc
gc_log = []
_t = [0.0]


def _gc_cb(phase, info):
    if phase == 'start':
        _t[0] = time.perf_counter()
    else:
        gc_log.append((info['generation'], 1e3 * (time.perf_counter() - _t[0]), info['collected']))


gc.callbacks.append(_gc_cb)
if os.environ.get('FEED_TRACE_NOGC') == '1':
    gc.collect(); gc.disable()

for rep in range(reps):
    src = iter(feed.BatchFeed(epoch(steps), dev, hp.MAX_TRAIN_LEN, mode=mode))
    evs, t_next, t_step = [], [], []
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
    while True:
        a = time.perf_counter()
        try:
            x = next(src)
        except StopIteration:
            break
        b = time.perf_counter()
        model.train_step(x)
        c = time.perf_counter()
        e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
        t_next.append(1e3 * (b - a)); t_step.append(1e3 * (c - b))
    torch.cuda.synchronize()
    dt = 1e3 * (time.perf_counter() - t0) / steps
    gpu = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(len(evs) - 1)])
    tn, ts = np.array(t_next), np.array(t_step)
    print('rep %d mode %s: %.3f ms/step | gpu step ms: median %.3f mean %.3f p90 %.3f max %.3f | host next(): median %.3f '
          'p90 %.3f | host train_step: median %.3f p90 %.3f' % (
              rep, mode, dt, np.median(gpu), gpu.mean(), np.percentile(gpu, 90), gpu.max(), np.median(tn),
              np.percentile(tn, 90), np.median(ts), np.percentile(ts, 90)), flush=True)
    worst = int(np.argmax(gpu))
    print('   worst gpu step #%d: gpu %.2f ms, host next %.2f, host step %.2f (neighbours gpu %s)' % (
        worst, gpu[worst], tn[worst], ts[worst], ' '.join('%.2f' % v for v in gpu[max(0, worst - 2):worst + 3])))
    hw = int(np.argmax(tn + ts))
    print('   worst host iteration #%d: next %.2f + step %.2f ms (gpu %.2f)' % (hw, tn[hw], ts[hw], gpu[hw]))
    print('   gc during rep: %s' % [(g, round(ms, 2), n) for g, ms, n in gc_log if ms > 0.5][-8:])
    del gc_log[:]
    print('   gpu[10:34] ' + ' '.join('%.2f' % v for v in gpu[10:34]), flush=True)
    print('   next[10:34] ' + ' '.join('%.2f' % v for v in tn[10:34]), flush=True)
    print('   step[10:34] ' + ' '.join('%.2f' % v for v in ts[10:34]), flush=True)
assert ops.lstm_status_ok()
