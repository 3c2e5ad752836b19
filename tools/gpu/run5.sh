cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
S=$(date +%s)
python bench.py > gpurun_out/default.json 2> gpurun_out/default.err; echo "rc=$? wall=$(( $(date +%s) - S )) s"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/default.json').read().strip().splitlines()[-1])
print(d['ms_per_step'], d['parity_ok'], d['roofline']['frac'], d['parity'].get('x6_vs_exact'), d.get('e2e',{}).get('ms_per_step'))
for k,v in d.get('also',{}).items():
    print(k, v['ms_per_step'], v['value'], v.get('parity_ok'), v['roofline'].get('frac'), v['wall_s'], (v.get('cpu_baseline') or {}).get('value'))
PY
tail -3 gpurun_out/default.err | cut -c1-400
timeout 900 python -m pytest tests/test_gpu_trained_parity.py "tests/test_gpu_round4.py::test_bench_two_ranks_functional" -x -q -s 2>&1 | grep -v "^TRAINED-PARITY\|^$" | tail -8
