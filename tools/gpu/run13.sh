cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_lstm.py tests/test_gpu_properties.py -x -q 2>&1 | tail -2
timeout 900 python -m pytest tests/test_gpu_model.py -x -q -k "h600 or 4x600" 2>&1 | tail -2
for c in cfg4h600 cfg2; do python bench.py --config $c --steps 20 --warmup 5 --no-cpu-baseline --no-e2e 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$c', d['ms_per_step'], d.get('parity_ok'), d['roofline'].get('lstm_fwd_us'), d['roofline'].get('lstm_bwd_us'), d['roofline'].get('frac'))
"; done
