# scratch script for tools/gpu/grun.sh (edit freely)
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_gpu_gemm_x6.py tests/test_gpu_gemm.py -x -q 2>&1 | tail -3
echo "--- TN, slices summed by the last arriver (default)"; python tools/bench_gemm_x6_tn.py 2>&1 | grep -v amdgpu.ids
echo "--- TN, slices summed by a second launch (round 5)"; DANET_EXPERT="gemm_x6_plan=262144" python tools/bench_gemm_x6_tn.py 2>&1 | grep -v amdgpu.ids
echo "--- NT hybrid"; python tools/bench_gemm_x6_hybrid.py 2>&1 | grep -v amdgpu.ids
b() { DANET_EXPERT="$2" python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-e2e --no-parity-check --no-also 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$1', d['ms_per_step'], d['roofline'].get('lstm_fwd_us'), d['roofline'].get('lstm_bwd_us'))
"; }
for i in 1 2; do
b "default              " ""
b "tn-two-launch        " "gemm_x6_plan=262144"
b "nt-plain             " "gemm_x6_plan=131072"
b "tn-two-launch+plain  " "gemm_x6_plan=393216"
b "bwd_s=2              " "lstm_bwd_s=2"
done
