cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
DANET_LSTM_FWD_FUSED=1 python tools/trace_lstm.py > gpurun_out/r05/r05_d_lstm_phase_trace.txt 2>&1
run() { tag=$1; shift; env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())
print('$tag', 'ms', d['ms_per_step'], 'steps', d['train_steps_before_mask_check'], 'e2e epoch mean loss', round(d['e2e']['epoch_mean_loss'],1), 'mask err vs f64', '%.2e' % d['mask_max_abs_err_vs_oracle'], 'f32 oracle', '%.2e' % d['mask_err_f32_oracle'], 'embed', '%.2e' % d['parity']['embed']['hip'])
"; }
run "default (x6 products, fused forward)          " A=1
run "exact fp32, default schedules                 " DANET_GEMM_X6=0 DANET_LSTM_FWD_FUSED=0
run "exact fp32, tile GEMMs instead of stream-K    " DANET_GEMM_X6=0 DANET_LSTM_FWD_FUSED=0 DANET_EXPERT=streamk=0
run "exact fp32, four split-K launches per dW group" DANET_GEMM_X6=0 DANET_LSTM_FWD_FUSED=0 DANET_EXPERT=grouped_dw=0
run "exact fp32, BPTT geometry U=32                " DANET_GEMM_X6=0 DANET_LSTM_FWD_FUSED=0 DANET_EXPERT=lstm_bwd_u=32
run "x6 products, unfused exact-fp32 forward       " DANET_LSTM_FWD_FUSED=0
