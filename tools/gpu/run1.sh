cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_round2.py tests/test_gpu_parity.py tests/test_gpu_properties.py -x -q 2>&1 | tail -6
timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-e2e 2>gpurun_out/b.err | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print(d['ms_per_step'], d.get('parity_ok'), d['roofline'].get('lstm_fwd_us'), d['roofline'].get('lstm_bwd_us'), d.get('mask_max_abs_err_vs_oracle'), d.get('mask_err_f32_oracle'))
"
tail -2 gpurun_out/b.err | cut -c1-300
