# BPTT experiments (GPU): twin co-location on one XCD; earlier: CU exclusivity (negative)
B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check"
run() { tag=$1; shift; env "$@" $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$tag', d['ms_per_step'], 'bwd %.1f fwd %.1f group %.1f' % (k['lstm_bwd']['avg_us'], k['lstm_fwd']['avg_us'], k['gemm_f32_group']['avg_us']))" >> gpurun_out/exp1.log 2>&1; }
rm -f gpurun_out/exp1.log
run default A=1
run twin_xcd DANET_LSTM_BWD_TWIN_XCD=1
run default2 A=1
run twin_xcd2 DANET_LSTM_BWD_TWIN_XCD=1
