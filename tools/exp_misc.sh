B="python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity-check"
run() { tag=$1; shift; env "$@" $B 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$tag', d['ms_per_step'], {n:v['avg_us'] for n,v in k.items() if n.startswith('gemm_f32')}, 'bwd %.1f' % (k['lstm_bwd']['avg_us']))" >> gpurun_out/exp9.log 2>&1; }
rm -f gpurun_out/exp9.log
run lb3_wgs512 A=1
run lb3_wgs768 DANET_GEMM_WGS=768
run lb3_wgs768_sk7 DANET_GEMM_WGS=768 DANET_STREAMK=7
run lb3_wgs512 A=1
run lb3_wgs768 DANET_GEMM_WGS=768
