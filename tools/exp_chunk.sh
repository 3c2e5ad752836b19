#!/bin/bash
# A/B of library build variants (csrc/libdanet_hip_<v>.so via DANET_LIB_PATH): head kernels + step
for v in "$@"; do
  if [ "$v" = default ]; then unset DANET_LIB_PATH; else export DANET_LIB_PATH=$PWD/danet-tensorflow_amd/csrc/libdanet_hip_$v.so; fi
  echo "== $v"
  python tools/bench_heads_fused.py 2>/dev/null
  python tools/bench_heads_fused.py --cfg4 2>/dev/null | tail -5
  for i in 1 2; do python bench.py --no-parity-check --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c 'import json,sys; print("cfg2 ms/step", json.loads(sys.stdin.read())["ms_per_step"])'; done
  for i in 1 2; do python bench.py --config cfg4 --no-parity-check --no-cpu-baseline --steps 100 --warmup 20 2>/dev/null | tail -1 | python -c 'import json,sys; print("cfg4 ms/step", json.loads(sys.stdin.read())["ms_per_step"])'; done
done
