cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_gemm_x6.py tests/test_gpu_gemm.py tests/test_gpu_fullsize.py tests/test_gpu_model.py tests/test_gpu_dist.py -x -q 2>&1 | tail -3
timeout 300 python tools/bench_gemm_x6_nt.py 2>&1 | grep -v amdgpu.ids | cut -c1-90
b() { python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-e2e --no-parity-check 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$1', d['ms_per_step'], d['roofline'].get('lstm_fwd_us'), d['roofline'].get('lstm_bwd_us'))
"; }
b new; b new
