for y in 0 8 16 24 32; do
  for r in 1 2; do
  ms=$(DANET_GEMM_YIELD=$y python bench.py --no-parity-check --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], d["roofline"]["lstm_bwd_us"], d["kernels"]["gemm_f32_group"]["avg_us"])')
  echo "yield $y: $ms"
  done
done
