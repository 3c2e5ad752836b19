'''
Does the one-off 13-19 ms recurrent-kernel stall still occur?  (GPU)

Runs `bench.py --step-times` N times in fresh processes (the stall was once per process)
with the given settle-step count and prints, per run, ms/step and the largest single-step
GPU time; a run is "stalled" when one step exceeds 2x the median step.
    python tools/stall_sweep.py N [settle_steps] [max_steps_in_flight]
'''
import json
import os
import re
import statistics
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    env = dict(os.environ)
    if len(sys.argv) > 2:
        env['DANET_BENCH_SETTLE_STEPS'] = sys.argv[2]
    if len(sys.argv) > 3:
        env['DANET_MAX_STEPS_IN_FLIGHT'] = sys.argv[3]
    stalled, rows = 0, []
    for i in range(n):
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '20', '--warmup', '5',
                            '--no-cpu-baseline', '--no-parity-check', '--step-times'],
                           capture_output=True, text=True, env=env)
        line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
        ms = json.loads(line)['ms_per_step']
        m = re.search(r'per-step GPU ms: (.*)', r.stderr)
        steps = [float(x) for x in m.group(1).split()]
        med = statistics.median(steps)
        bad = max(steps) > 2 * med
        stalled += bad
        rows.append((ms, max(steps), med))
        print('run %2d: %.3f ms/step, median step %.3f, max step %.3f%s' %
              (i, ms, med, max(steps), '  <-- STALL' if bad else ''), flush=True)
    print('settle=%s in_flight=%s: %d of %d runs stalled; ms/step min %.3f median %.3f max %.3f' % (
        env.get('DANET_BENCH_SETTLE_STEPS', 'default'), env.get('DANET_MAX_STEPS_IN_FLIGHT', 'default'),
        stalled, n, min(r[0] for r in rows), statistics.median(r[0] for r in rows), max(r[0] for r in rows)))


if __name__ == '__main__':
    main()
