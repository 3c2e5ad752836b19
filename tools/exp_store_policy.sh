#!/bin/bash
# Cache policy of the BPTT publish stores (build variants aux17 = sc0 sc1, aux18 = sc1 nt; default sc1):
# hand-off test, step time, PMC FETCH/WRITE per BPTT launch.
for v in "" _aux17 _aux18; do
  export DANET_LIB_PATH=$PWD/danet-tensorflow_amd/csrc/libdanet_hip$v.so
  echo "== policy${v:-_sc1}"
  python -m pytest tests -x -q -m gpu -k "handoff_under_uneven_load or lstm_layer_fwd_bwd" 2>&1 | tail -1
  for r in 1 2; do python bench.py --no-parity-check --no-cpu-baseline --steps 200 --warmup 20 2>/dev/null | tail -1 | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("ms/step", d["ms_per_step"], "bwd call us", d["roofline"]["lstm_bwd_us"])'; done
  bash profiles/run_pmc.sh sp$v cfg2 x > /dev/null 2>&1
  python - "gpurun_out/pmc_sp$v/summary.json" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    if 'lstm_bwd_rs' in k:
        print(k, 'FETCH %.1f MB  WRITE %.1f MB per launch' % (v['FETCH_SIZE']['per_launch'] * 1024 / 1e6, v['WRITE_SIZE']['per_launch'] * 1024 / 1e6))
PY
done
