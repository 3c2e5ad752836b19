#!/bin/bash
# Round-6 evidence in one go (GPU box):  bash profiles/run_r06.sh <tag> <git-rev>
# -> gpurun_out/r06/<tag>_*: un-profiled bench lines of every config, rocprofv3 kernel stats +
# the lines those profiled runs printed, one step's kernel timeline, PMC traffic (cfg2, cfg4h600),
# MFMA-busy, the GEMM shapes.
set -u
TAG=${1:-x}; REV=${2:-unknown}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r06
mkdir -p "$OUT"
cd "$ROOT"
# the driver's command first (default: cfg 2 + the short cfg4h600 / cfg5 sub-records), then every config at full length
python bench.py 2> "$OUT/${TAG}_bench_default.err" | tail -1 > "$OUT/${TAG}_bench_default.json"
for c in cfg2 cfg4 cfg4h600 cfg5 cfg5-kmeans; do
  python bench.py --config $c --steps 20 --warmup 5 --no-also 2> "$OUT/${TAG}_bench_$c.err" | tail -1 > "$OUT/${TAG}_bench_$c.json"
done
DANET_FORCE_DIST=1 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_cfg2_rccl1.json"
for c in cfg2 cfg4 cfg4h600 cfg5; do
  bash profiles/run_rocprof.sh ${TAG}_$c --config $c > "$OUT/${TAG}_rocprof_$c.log" 2>&1
  cp gpurun_out/prof_${TAG}_$c/kernel_stats.csv "$OUT/${TAG}_kernel_stats_$c.csv" 2>/dev/null
  grep -E '^\{' gpurun_out/prof_${TAG}_$c/bench.log | tail -1 > "$OUT/${TAG}_bench_under_rocprof_$c.json"
done
bash tools/timeline.sh ${TAG} > /dev/null 2>&1
cp gpurun_out/${TAG}_timeline.txt "$OUT/${TAG}_step_timeline.txt" 2>/dev/null
bash profiles/run_pmc.sh ${TAG}_cfg2 cfg2 $REV > "$OUT/${TAG}_pmc_cfg2.log" 2>&1
cp gpurun_out/pmc_${TAG}_cfg2/summary.json "$OUT/${TAG}_pmc_summary.json" 2>/dev/null
bash profiles/run_pmc.sh ${TAG}_cfg4h600 cfg4h600 $REV > "$OUT/${TAG}_pmc_cfg4h600.log" 2>&1
cp gpurun_out/pmc_${TAG}_cfg4h600/summary.json "$OUT/${TAG}_pmc_summary_cfg4h600.json" 2>/dev/null
bash profiles/run_pmc_mfma.sh ${TAG} > "$OUT/${TAG}_pmc_mfma.log" 2>&1
cp gpurun_out/pmc_mfma_${TAG}/summary.json "$OUT/${TAG}_pmc_mfma_summary.json" 2>/dev/null
python tools/bench_gemm.py > "$OUT/${TAG}_gemm_shapes.txt" 2>&1
python tools/bench_gemm_x6_nt.py > "$OUT/${TAG}_gemm_x6_nt.txt" 2>&1
python tools/bench_gemm_x6_tn.py > "$OUT/${TAG}_gemm_x6_tn.txt" 2>&1
python tools/bptt_neighbour_probe.py > "$OUT/${TAG}_bptt_neighbour.txt" 2>&1
# every product on the exact-fp32 matrix instructions (x6 off, hoisted input half + unfused forward): the same
# step count before the mask check as the default line (e2e on)
DANET_GEMM_X6=0 DANET_LSTM_FWD_FUSED=0 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-also 2>/dev/null | tail -1 > "$OUT/${TAG}_bench_cfg2_exact_fp32.json"
DANET_LSTM_FWD_FUSED=1 python tools/trace_lstm.py > "$OUT/${TAG}_lstm_phase_trace.txt" 2>&1
python tools/lstm_modes.py > "$OUT/${TAG}_lstm_bwd_modes.txt" 2>&1
python tools/feed_probe.py 100 > "$OUT/${TAG}_feed_probe.txt" 2>&1
python tools/feed_trace.py ahead 100 3 > "$OUT/${TAG}_feed_trace.txt" 2>&1
ls -la "$OUT"
# round 6: the hybrid NT schedule, the heads chain with the gradient partials in the forward, the heads' PMC
python tools/bench_gemm_x6_hybrid.py > "$OUT/${TAG}_gemm_x6_hybrid.txt" 2>&1
python tools/bench_heads_fused.py > "$OUT/${TAG}_heads_fused.txt" 2>&1
bash tools/heads_stats.sh > "$OUT/${TAG}_heads_stats.txt" 2>&1
bash tools/heads_pmc.sh > "$OUT/${TAG}_heads_pmc.txt" 2>&1
ls -la "$OUT"
