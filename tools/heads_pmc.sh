#!/bin/bash
# Usage (GPU box): bash tools/heads_pmc.sh  -- instruction mix of the estimator / separator / loss kernels
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
for C in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU" "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM" "GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum"; do
  rm -rf /tmp/hp
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/hp -o hp -- python $ROOT/tools/bench_heads_fused.py "$@" > /tmp/hp.log 2>&1 < /dev/null
  f=$(find /tmp/hp -name "*counter_collection.csv" | head -1)
  python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name'].split('(')[0][-40:]
    if not any(t in k for t in ('anchor_fwd', 'anchor_sep_bwd', 'sep_pit_bwd', 'sep_pit_fwd', 'truth_')): continue
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    n[(k, r['Counter_Name'])] += 1
for k, v in acc.items():
    print('%-42s' % k, '  '.join('%s %.3g' % (c, x / n[(k, c)]) for c, x in sorted(v.items())))
PY
done
