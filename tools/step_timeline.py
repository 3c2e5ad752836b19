#!/usr/bin/env python
'''Timeline of ONE train step from a rocprofv3 kernel trace (GPU box):

    cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- \
        python $REPO/bench.py --steps 6 --warmup 3 --no-cpu-baseline
    python tools/step_timeline.py /tmp/tl/.../tl_kernel_trace.csv [step_index [v]]

(bench.py records HIP events only on timed steps 0, 4, 8, ...: pick an un-instrumented one,
e.g. step_index = warmup + 5.)  Prints, for the chosen step (frontend_kernel to frontend_kernel), every kernel
with start offset / duration / queue, the busy time of the union of all kernels and the
idle gaps (no kernel running at all).'''
import csv
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    ev = []
    for r in rows:
        ev.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:44],
                   r.get('Queue_Id', '?')))
    ev.sort()
    # a step starts at its front-end kernel (the optimizer update is issued in pieces, the
    # first of them in the middle of backward, so adam_clip no longer delimits steps)
    first = [i for i, e in enumerate(ev) if e[2].startswith('frontend_kernel')]
    k = int(sys.argv[2]) if len(sys.argv) > 2 else len(first) - 2     # which step
    i0, i1 = first[k], first[k + 1]
    step = ev[i0:i1]
    t0 = step[0][0]
    print('step: %d kernels, %.1f us wall' % (len(step), (step[-1][1] - t0) / 1e3))
    busy, cur_end, gaps = 0, t0, []
    for s, e, n, q in step:
        if s > cur_end:
            gaps.append((cur_end - t0, s - cur_end, n))
            busy += e - s
            cur_end = e
        elif e > cur_end:
            busy += e - cur_end
            cur_end = e
    print('GPU busy (union) %.1f us; idle %.1f us in %d gaps' % (
        busy / 1e3, sum(g[1] for g in gaps) / 1e3, len(gaps)))
    verbose = len(sys.argv) > 3
    for s, e, n, q in step:
        if verbose or (e - s) > 20000:
            print('%9.1f %8.1f  q%-3s %s' % ((s - t0) / 1e3, (e - s) / 1e3, q, n))
    print('largest gaps (offset us, gap us, next kernel):')
    for g in sorted(gaps, key=lambda g: -g[1])[:12]:
        print('   %9.1f %7.1f  %s' % (g[0] / 1e3, g[1] / 1e3, g[2]))


if __name__ == '__main__':
    main()
