B="python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity-check"
run() { tag=$1; shift; env "$@" $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$tag', d['ms_per_step'], {n:v['avg_us'] for n,v in k.items() if n.startswith('gemm_f32:')}, 'bwd %.1f fwd %.1f group %.1f' % (k['lstm_bwd']['avg_us'], k['lstm_fwd']['avg_us'], k['gemm_f32_group']['avg_us']))" >> gpurun_out/exp2.log 2>&1; }
rm -f gpurun_out/exp2.log
run default A=1
run sk1 DANET_STREAMK=1
run sk2 DANET_STREAMK=2
run sk4 DANET_STREAMK=4
run sk7 DANET_STREAMK=7
run sk7_wgs256 DANET_STREAMK=7 DANET_GEMM_WGS=256
run sk7_wgs768 DANET_STREAMK=7 DANET_GEMM_WGS=768
run sk6 DANET_STREAMK=6
run default2 A=1
