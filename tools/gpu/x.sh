cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_optimizer.py tests/test_gpu_cli.py -x -q 2>&1 | tail -5
b() { python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-e2e --no-parity-check $2 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$1', d['ms_per_step'], d['roofline'].get('lstm_fwd_us'), d['roofline'].get('lstm_bwd_us'))
"; }
b defer; b inorder --no-defer-tail; b defer; b inorder --no-defer-tail;  b defer; b inorder --no-defer-tail
python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-parity-check 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('with e2e', d['ms_per_step'], d['e2e']['ms_per_step'], d['e2e']['sync_feed_ms_per_step'])
"
bash tools/timeline.sh defer > /dev/null 2>&1
