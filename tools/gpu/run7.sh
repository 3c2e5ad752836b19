cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
DANET_LSTM_FWD_FUSED=1 timeout 300 python tools/trace_lstm.py 2>&1 | sed -n 2,40p | grep -v "barrier\|to_next"
timeout 2000 python -m pytest tests/test_gpu_lstm.py tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_gpu_model.py -x -q 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-e2e --no-also 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print(d['ms_per_step'], d.get('parity_ok'), d['roofline'].get('lstm_fwd_us'), d['roofline'].get('lstm_bwd_us'), d['roofline']['frac'])
"; done
