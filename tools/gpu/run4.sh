cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
DANET_LSTM_FWD_FUSED=1 timeout 300 python tools/trace_lstm.py 2>&1 | sed -n 2,17p
timeout 900 python -m pytest tests/test_gpu_lstm.py tests/test_gpu_properties.py tests/test_gpu_parity.py -x -q 2>&1 | tail -2
timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-e2e 2>gpurun_out/b.err | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print(d['ms_per_step'], d.get('parity_ok'), d['roofline'].get('lstm_fwd_us'), d['roofline'].get('lstm_bwd_us'), d.get('mask_max_abs_err_vs_oracle'), d.get('mask_err_f32_oracle'), d['parity']['embed'])
"
