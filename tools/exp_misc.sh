B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity-check"
run() { tag=$1; shift; env "$@" $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$tag', d['ms_per_step'], 'bwd %.1f fwd %.1f group %.1f' % (k['lstm_bwd']['avg_us'], k['lstm_fwd']['avg_us'], k['gemm_f32_group']['avg_us']))" >> gpurun_out/exp11.log 2>&1; }
rm -f gpurun_out/exp11.log
run recompute A=1
run two_pass DANET_HEADS_RECOMPUTE=0
run recompute A=1
run two_pass DANET_HEADS_RECOMPUTE=0
