cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_lstm.py tests/test_gpu_parity.py tests/test_gpu_properties.py tests/test_gpu_fullsize.py tests/test_gpu_runtime.py -x -q 2>&1 | tail -3
b() { python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-e2e --no-parity-check $2 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$1', d['ms_per_step'], d['roofline'].get('lstm_fwd_us'), d['roofline'].get('lstm_bwd_us'))
"; }
P=$GRAFT_REPO_ROOT/danet-tensorflow_amd/csrc/libdanet_hip_prev.so
b cfg4h600 "--config cfg4h600"; DANET_LIB_PATH=$P b cfg4h600prev "--config cfg4h600"; b cfg4h600 "--config cfg4h600"; DANET_LIB_PATH=$P b cfg4h600prev "--config cfg4h600"
b cfg2; DANET_LIB_PATH=$P b cfg2prev
