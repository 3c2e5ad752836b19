#!/bin/bash
# One train step's kernel timeline under rocprofv3 (GPU box):
#   bash tools/timeline.sh <tag> [bench args...]     (environment overrides are inherited)
# -> gpurun_out/<tag>_timeline.txt, gpurun_out/<tag>_kernel_stats.csv
set -u
TAG=${1:-tl}; shift || true
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$ROOT/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl_$TAG; mkdir -p /tmp/tl_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tl_$TAG -o $TAG -- \
    python $ROOT/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-parity-check "$@" \
    > /tmp/tl_$TAG/bench.log 2>&1 < /dev/null
TRACE=$(find /tmp/tl_$TAG -name "*kernel_trace.csv" < /dev/null | head -1)
STATS=$(find /tmp/tl_$TAG -name "*kernel_stats.csv" < /dev/null | head -1)
[ -n "$STATS" ] && cp "$STATS" "$ROOT/gpurun_out/${TAG}_kernel_stats.csv"
if [ -n "$TRACE" ]; then
  python $ROOT/tools/step_timeline.py "$TRACE" 9 v > "$ROOT/gpurun_out/${TAG}_timeline.txt" 2>&1
else
  echo "no kernel trace" > "$ROOT/gpurun_out/${TAG}_timeline.txt"; tail -20 /tmp/tl_$TAG/bench.log >> "$ROOT/gpurun_out/${TAG}_timeline.txt"
fi
grep -E '^\{' /tmp/tl_$TAG/bench.log | cut -c1-200
