run() { tag=$1; cfg=$2; shift; shift; env "$@" python bench.py --config $cfg --steps 16 --warmup 4 --no-cpu-baseline --no-parity-check 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$tag', '$cfg', d['ms_per_step'], d['value'])" >> gpurun_out/exp12.log 2>&1; }
rm -f gpurun_out/exp12.log
run recompute cfg4 A=1
run two_pass cfg4 DANET_HEADS_RECOMPUTE=0
run recompute cfg4 A=1
run two_pass cfg4 DANET_HEADS_RECOMPUTE=0
run recompute cfg4h600 A=1
