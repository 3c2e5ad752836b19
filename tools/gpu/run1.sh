cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_round4.py tests/test_gpu_gemm_x6.py tests/test_gpu_parity.py -x -q 2>&1 | tail -4
timeout 300 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-e2e 2>/dev/null | cut -c1-160
