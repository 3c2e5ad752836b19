cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
python bench.py > gpurun_out/r05_n_bench_default.json 2> gpurun_out/r05_n_bench_default.err; echo rc=$?
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r05_n_bench_default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['value'], d['parity_ok'], d['roofline']['frac'], d['roofline']['lstm_fwd_us'], d['roofline']['lstm_bwd_us'], d['e2e']['ms_per_step'], d['roofline']['gemm_f32']['frac'])
for k,v in d.get('also',{}).items():
    print(k, v['ms_per_step'], v['value'], v.get('parity_ok'), v['roofline'].get('frac'), v['wall_s'])
PY
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
