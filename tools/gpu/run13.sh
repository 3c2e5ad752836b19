cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for i in 1 2; do for c in cfg2; do python bench.py --config $c --steps 60 --warmup 10 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$c', d['ms_per_step'], d.get('parity_ok'), d['roofline'].get('lstm_fwd_us'), d['roofline'].get('lstm_bwd_us'), d['roofline'].get('frac'))
"; done; done
python tools/bptt_neighbour_probe.py 2>&1 | sed -n 2,4p
