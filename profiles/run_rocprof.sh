#!/bin/bash
# Usage (on the GPU box, via gpurun): bash profiles/run_rocprof.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>/{kernel_stats.csv,bench.log}; stdin-safe, bounded.
set -u
TAG=${1:-r1}; shift || true
OUT=${GRAFT_REPO_ROOT:-$(pwd)}/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o $TAG -- \
    python ${GRAFT_REPO_ROOT:-/root/repo}/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity-check "$@" \
    > "$OUT/bench.log" 2>&1 < /dev/null
find /tmp/prof_$TAG -type f < /dev/null | head -20
for f in $(find /tmp/prof_$TAG -name "*kernel_stats.csv" < /dev/null); do cp "$f" "$OUT/kernel_stats.csv"; done
if [ -f "$OUT/kernel_stats.csv" ]; then head -25 "$OUT/kernel_stats.csv" | cut -c1-200; else echo "no kernel_stats.csv"; fi
grep -E '^\{' "$OUT/bench.log" | cut -c1-300
