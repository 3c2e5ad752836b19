B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity-check"
run() { tag=$1; shift; env "$@" $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$tag', d['ms_per_step'], 'bwd %.1f fwd %.1f group %.1f' % (k['lstm_bwd']['avg_us'], k['lstm_fwd']['avg_us'], k['gemm_f32_group']['avg_us']))" >> gpurun_out/exp10.log 2>&1; }
rm -f gpurun_out/exp10.log
run default A=1
run h DANET_LSTM_BWD_FUSED=h
run h_y8 DANET_LSTM_BWD_FUSED=h DANET_GEMM_YIELD=8
run h_y0 DANET_LSTM_BWD_FUSED=h DANET_GEMM_YIELD=0
run h_y32 DANET_LSTM_BWD_FUSED=h DANET_GEMM_YIELD=32
run default A=1
run h DANET_LSTM_BWD_FUSED=h
