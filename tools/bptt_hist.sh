# GPU box: distribution of the in-step BPTT kernel durations (rocprofv3 kernel trace of bench.py)
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/prof_h
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/prof_h -o h -- python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-parity-check --no-e2e > /dev/null 2>&1 < /dev/null
f=$(find /tmp/prof_h -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
d=[(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3 for r in rows if 'lstm_bwd_rs' in r['Kernel_Name']]
d=d[-3*40:]
import statistics
print('n',len(d),'mean %.1f median %.1f min %.1f max %.1f'%(statistics.mean(d),statistics.median(d),min(d),max(d)))
h=collections.Counter(int(x//10)*10 for x in d)
for k in sorted(h): print(k, h[k])
print('positions of launches > 420 us (index from the end, 3 per step):', [len(d)-i for i,x in enumerate(d) if x>420])
PY
