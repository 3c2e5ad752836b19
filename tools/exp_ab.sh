#!/bin/bash
# Same-box A/B of run-time switches: alternates the variants R times and prints ms/step of each run.
# usage: tools/exp_ab.sh "<env A>" "<env B>" [rounds] [extra bench args]
A="$1"; B="$2"; R="${3:-4}"; shift 3 || true
mkdir -p gpurun_out
for r in $(seq $R); do
  for v in A B; do
    if [ $v = A ]; then E="$A"; else E="$B"; fi
    ms=$(env $E python bench.py --no-parity-check --no-cpu-baseline --steps 200 --warmup 20 "$@" 2>/dev/null | tail -1 | python -c 'import json,sys; print(json.loads(sys.stdin.read())["ms_per_step"])')
    echo "$r $v [$E] $ms"
  done
done | tee gpurun_out/exp_ab.txt
