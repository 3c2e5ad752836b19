#!/bin/bash
# usage: tools/gpu/grun.sh <timeout_s> <script under tools/gpu/> -- retries while no GPU slot is free
T=$1; S=$2
for i in $(seq 1 40); do
  out=$(${GPURUN:-/usr/local/graft/bin/gpurun} --timeout $T -- "bash tools/gpu/$S" 2>&1)
  if echo "$out" | grep -q "status=transient\|nothing was charged"; then sleep 90; continue; fi
  echo "$out"; exit 0
done
echo "no GPU slot after 40 tries"; exit 3
