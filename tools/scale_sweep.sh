#!/bin/bash
# One command for the first multi-GPU lease: N in {1,2,4,8} x gradient-reduction schedule in
# {0 (ONE all-reduce per step, the default), tail, 1}; one bench JSON line per cell under
# gpurun_out/scale/ plus a table.  bench.py launches its own ranks (torch.distributed.run on
# 127.0.0.1, one process per GPU, RCCL over xGMI).
#   bash tools/scale_sweep.sh [steps] [warmup]
set -u
cd "$(dirname "$0")/.."
STEPS=${1:-30}; WARM=${2:-5}
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out/scale
NG=$(python -c 'import torch; print(torch.cuda.device_count())')
for N in 1 2 4 8; do
  [ "$N" -le "$NG" ] || { echo "skip N=$N (only $NG GPUs visible)"; continue; }
  for SCHED in 0 tail 1; do
    OUT=gpurun_out/scale/cfg2_n${N}_sched${SCHED}.json
    EXTRA=""
    [ "$N" -gt 1 ] && EXTRA="--no-cpu-baseline"
    # N = 1 takes the same spawn path (a 1-rank RCCL group) so that the curve starts from a
    # run that includes the collective's launch cost
    DANET_FORCE_DIST=1 timeout 900 python bench.py --gpus $N --steps $STEPS --warmup $WARM \
        --allreduce-schedule $SCHED --no-cpu-baseline --no-parity-check $EXTRA \
        2> gpurun_out/scale/cfg2_n${N}_sched${SCHED}.err | tail -1 > $OUT
    python - "$OUT" "$N" "$SCHED" <<'PY'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read())
    print('N=%s sched=%-4s %8.1f mixture-s/s  %.3f ms/step  collectives/step=%s  rccl_ranks=%s  allreduce alone %.3f ms'
          % (sys.argv[2], sys.argv[3], d['value'], d['ms_per_step'], d['config'].get('collectives_per_step'),
             d['rccl_ranks'], d.get('allreduce_ms_standalone') or float('nan')))
except Exception as e:
    print('N=%s sched=%s FAILED: %s' % (sys.argv[2], sys.argv[3], e))
PY
  done
done
