run() { tag=$1; cfg=$2; shift; shift; env "$@" python bench.py --config $cfg --steps 12 --warmup 3 --no-cpu-baseline --no-parity-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$tag', '$cfg', d['ms_per_step'], d['value'], 'fwd %.1f' % (k['lstm_fwd']['avg_us']), d['roofline'].get('us_per_timestep'))" >> gpurun_out/exp5.log 2>&1; }
rm -f gpurun_out/exp5.log
run xl cfg5 A=1
run noxl cfg5 DANET_LSTM_FWD_SMALL_XL=0
run xl cfg5 A=1
run noxl cfg5 DANET_LSTM_FWD_SMALL_XL=0
run xl cfg5-kmeans A=1
