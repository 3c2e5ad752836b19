#!/bin/bash
# Usage (GPU box): bash profiles/run_pmc_mfma.sh <tag>
# One rocprofv3 --pmc pass (kernel-trace only) with SQ_VALU_MFMA_BUSY_CYCLES and
# GRBM_GUI_ACTIVE.  GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (one GRBM each), the SQ
# counter summed over all SIMDs, so per kernel:
#   effective clock  = GUI_ACTIVE / 8 / kernel duration          (2.37-2.45 GHz measured)
#   MFMA utilisation = MFMA_BUSY / (GUI_ACTIVE / 8 * 256 CUs * 4 SIMDs)
# (checks against the flop rate: dX/dYc at 86 TFLOP/s = 0.55 of 157.3 -> 0.56 here).
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/pmc_mfma_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmcm_$TAG
timeout 500 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
    -d /tmp/pmcm_$TAG -o m -- python $ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-parity-check \
    > "$OUT/run.log" 2>&1 < /dev/null
CC=$(find /tmp/pmcm_$TAG -name "*counter_collection.csv" < /dev/null | head -1)
KT=$(find /tmp/pmcm_$TAG -name "*kernel_trace.csv" < /dev/null | head -1)
python - "$CC" "$KT" "$OUT" <<'PY'
import csv, json, sys, collections, os
cc, kt, out = sys.argv[1:4]
dur = {}
if kt and os.path.exists(kt):
    for r in csv.DictReader(open(kt)):
        dur[r['Dispatch_Id']] = int(r['End_Timestamp']) - int(r['Start_Timestamp'])
acc = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
for r in csv.DictReader(open(cc)):
    k = r['Kernel_Name'].split('(')[0]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
        n[k] += 1
        acc[k]['ns'] += dur.get(r['Dispatch_Id'], 0)
res = {}
for k, v in acc.items():
    if not n[k] or not v.get('GRBM_GUI_ACTIVE'):
        continue
    res[k] = dict(launches=n[k], mfma_busy_cycles=v['SQ_VALU_MFMA_BUSY_CYCLES'] / n[k],
                  gui_active_cycles=v['GRBM_GUI_ACTIVE'] / n[k],
                  mfma_util=v['SQ_VALU_MFMA_BUSY_CYCLES'] / (v['GRBM_GUI_ACTIVE'] / 8 * 256 * 4),
                  clock_ghz=(v['GRBM_GUI_ACTIVE'] / 8 / v['ns']) if v['ns'] else None)
json.dump(res, open(os.path.join(out, 'summary.json'), 'w'), indent=1, sort_keys=True)
for k in sorted(res, key=lambda k: -res[k]['mfma_busy_cycles'] * res[k]['launches'])[:10]:
    r = res[k]
    print('%-46s n=%3d  mfma_util %.3f  clock %s GHz' % (k[:46], r['launches'], r['mfma_util'],
          ('%.2f' % r['clock_ghz']) if r['clock_ghz'] else '?'))
PY
