# scratch script for tools/gpu/grun.sh (edit freely): lstm / parity tests + two short bench lines
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_lstm.py tests/test_gpu_parity.py tests/test_gpu_properties.py -x -q 2>&1 | tail -2
b() { python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-e2e --no-parity-check 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$1', d['ms_per_step'], d['roofline'].get('lstm_fwd_us'), d['roofline'].get('lstm_bwd_us'))
"; }
b new; b new
