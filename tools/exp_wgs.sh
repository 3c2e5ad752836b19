#!/bin/bash
# stream-K grid size sweep (dYc, dX, the bottom layer's group): ms per step and us per tagged GEMM
for w in 512 480 320 448 512; do
  for r in 1 2; do
  DANET_GEMM_WGS=$w python bench.py --no-parity-check --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); k=d["kernels"]; print("wgs '$w':", d["ms_per_step"], "dYc", k["gemm_f32:dYc"]["avg_us"], "dX", k["gemm_f32:dX"]["avg_us"], "proj", k["gemm_f32:proj"]["avg_us"])'
  done
done
