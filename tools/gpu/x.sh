cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
b() { python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-e2e --no-parity-check $2 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$1', d['ms_per_step'], d['roofline'].get('lstm_fwd_us'), d['roofline'].get('lstm_bwd_us'))
"; }
P=$GRAFT_REPO_ROOT/danet-tensorflow_amd/csrc/libdanet_hip_prev.so
b new; DANET_LIB_PATH=$P b prev; b new; DANET_LIB_PATH=$P b prev
b cfg4h600 "--config cfg4h600"; DANET_LIB_PATH=$P b cfg4h600prev "--config cfg4h600"
b cfg4 "--config cfg4"; DANET_LIB_PATH=$P b cfg4prev "--config cfg4"
DANET_LSTM_FWD_FUSED=1 python tools/trace_lstm.py 2>&1 | grep -v amdgpu | head -40 > gpurun_out/trace_new.txt
