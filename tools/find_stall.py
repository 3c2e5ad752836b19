#!/usr/bin/env python
'''find_stall.py <kernel_trace.csv>: kernels longer than 3 ms and idle gaps longer than 3 ms in a
rocprofv3 kernel trace (looking for the sporadic 20-30 ms train step).'''
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0][:50], r.get('Queue_Id', '?')) for r in rows)
t0 = ev[0][0]
end = 0
hits = 0
for i, (s, e, n, q) in enumerate(ev):
    if e - s > 3e6:
        hits += 1
        print('LONG  %9.3f ms  dur %8.3f ms  q%s %s' % ((s - t0) / 1e6, (e - s) / 1e6, q, n))
        for s2, e2, n2, q2 in ev[max(0, i - 4):i + 6]:
            print('        %9.3f  +%8.3f ms  q%s %s' % ((s2 - t0) / 1e6, (e2 - s2) / 1e6, q2, n2))
    if end and s - end > 3e6:
        hits += 1
        print('GAP   %9.3f ms  gap %8.3f ms before q%s %s' % ((s - t0) / 1e6, (s - end) / 1e6, q, n))
    end = max(end, e)
print('%d kernels, %d hits' % (len(ev), hits))
