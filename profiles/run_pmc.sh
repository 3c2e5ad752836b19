#!/bin/bash
# Usage (on the GPU box): bash profiles/run_pmc.sh <tag> [bench config = cfg2] [git revision]
# Two separate rocprofv3 --pmc passes (FETCH_SIZE costs 3 of 4 TCC slots, WRITE_SIZE 2:
# they do not fit one pass), kernel-trace only.  Aggregates per kernel name into
# gpurun_out/pmc_<tag>/summary.json (per-launch averages, raw counter units).
set -u
TAG=${1:-r1}
CONFIG=${2:-cfg2}
GITREV=${3:-unknown}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_${TAG}_$C
  timeout 500 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc_${TAG}_$C -o $C -- \
      python $ROOT/bench.py --config $CONFIG --steps 4 --warmup 2 --no-cpu-baseline --no-parity-check > "$OUT/$C.log" 2>&1 < /dev/null
  for f in $(find /tmp/pmc_${TAG}_$C -name "*counter_collection.csv" < /dev/null); do cp "$f" "$OUT/$C.csv"; done
done
python - "$OUT" "$TAG" "$CONFIG" "$GITREV" <<'PY'
import csv, json, sys, collections, os
out = sys.argv[1]
res = collections.defaultdict(dict)
res['_meta'] = dict(tag=sys.argv[2], workload=sys.argv[3], git=sys.argv[4],
                    command='bench.py --config %s --steps 4 --warmup 2 --no-cpu-baseline' % sys.argv[3],
                    units='FETCH_SIZE / WRITE_SIZE in KB per launch (raw counter units)')
for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    p = os.path.join(out, c + '.csv')
    if not os.path.exists(p):
        continue
    acc = collections.defaultdict(lambda: [0, 0.0])
    with open(p) as f:
        for row in csv.DictReader(f):
            if row.get('Counter_Name') != c:
                continue
            k = row['Kernel_Name'].split('(')[0]
            acc[k][0] += 1
            acc[k][1] += float(row['Counter_Value'])
    for k, (n, v) in acc.items():
        res[k][c] = dict(launches=n, per_launch=v / n)
json.dump(res, open(os.path.join(out, 'summary.json'), 'w'), indent=1, sort_keys=True)
for k in sorted((k for k in res if not k.startswith('_')),
                key=lambda k: -res[k].get('FETCH_SIZE', {}).get('per_launch', 0))[:12]:
    print(k[:60], {c: round(v['per_launch'], 1) for c, v in res[k].items()})
PY
rm -f "$OUT/FETCH_SIZE.csv" "$OUT/WRITE_SIZE.csv"   # keep the summary only (size)
