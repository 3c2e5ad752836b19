#!/bin/bash
# Usage (GPU box): bash tools/gemm_pmc.sh  -- clock and MFMA utilisation of the GEMM kernels of tools/gemm_probe.py
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/gpmc
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv \
    -d /tmp/gpmc -o m -- python $ROOT/tools/gemm_probe.py > /tmp/gpmc.log 2>&1 < /dev/null
CC=$(find /tmp/gpmc -name "*counter_collection.csv" < /dev/null | head -1)
KT=$(find /tmp/gpmc -name "*kernel_trace.csv" < /dev/null | head -1)
python - "$CC" "$KT" <<'PY'
import csv, sys, collections
cc, kt = sys.argv[1:3]
dur = {}
for r in csv.DictReader(open(kt)):
    dur[r['Dispatch_Id']] = (int(r['End_Timestamp']) - int(r['Start_Timestamp']), r.get('Grid_Size', r.get('Grid_Size_X', '?')))
rows = collections.OrderedDict()
for r in csv.DictReader(open(cc)):
    d = rows.setdefault(r['Dispatch_Id'], {'k': r['Kernel_Name'][:60]})
    d[r['Counter_Name']] = float(r['Counter_Value'])
agg = collections.OrderedDict()
for i, d in rows.items():
    if 'GRBM_GUI_ACTIVE' not in d or i not in dur:
        continue
    ns, grid = dur[i]
    key = (d['k'], grid, round(ns / 20e3))
    a = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
    a[0] += 1; a[1] += ns; a[2] += d['GRBM_GUI_ACTIVE']; a[3] += d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0)
for (k, grid, _), (n, ns, gui, mf) in agg.items():
    if ns / n < 50e3:
        continue
    clk = gui / 8 / ns
    print('%-60s grid %8s n %3d  %8.1f us  clock %.2f GHz  MFMA busy %.1f %%' % (
        k, grid, n, ns / n / 1e3, clk, 100 * mf / (gui / 8 * 1024)))
PY
