run() { tag=$1; cfg=$2; shift; shift; env "$@" python bench.py --config $cfg --steps 12 --warmup 3 --no-cpu-baseline --no-parity-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels']
print('$tag', '$cfg', d['ms_per_step'], d['value'])" >> gpurun_out/exp7.log 2>&1; }
rm -f gpurun_out/exp7.log
run new cfg4 A=1
run new cfg4 A=1
run new cfg4h600 A=1
