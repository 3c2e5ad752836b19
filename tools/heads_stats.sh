#!/bin/bash
# Usage (GPU box): bash tools/heads_stats.sh [--cfg4]  -- per-kernel average of tools/bench_heads_fused.py (the train step's head kernels, through the C ABI) under rocprofv3
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/hs
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/hs -o hs -- python $ROOT/tools/bench_heads_fused.py "$@" > /tmp/hs.log 2>&1 < /dev/null
f=$(find /tmp/hs -name "*kernel_stats.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].split('(')[0][-44:]
    if any(k in n for k in ('anchor', 'separate', 'pit_', 'sum_chunks', 'truth')):
        print('%-46s calls %4s  avg %7.1f us' % (n, r['Calls'], float(r['AverageNs']) / 1e3))
PY
