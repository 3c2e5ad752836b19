cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_gemm_x6.py tests/test_gpu_gemm.py tests/test_gpu_lstm.py -x -q 2>&1 | tail -3
timeout 300 python tools/bench_gemm_x6_tn.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/x6_tn.txt
b() { python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-e2e --no-parity-check 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$1', d['ms_per_step'], d['roofline'].get('lstm_fwd_us'), d['roofline'].get('lstm_bwd_us'))
"; }
b new; b new
