cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
b() { python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-e2e --no-parity-check 2>/dev/null | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l)
    except Exception: continue
    print('$1', d['ms_per_step'], d['roofline'].get('lstm_fwd_us'), d['roofline'].get('lstm_bwd_us'))
"; }
for r in 1 2; do
b default
for n in 0 1 3 4; do DANET_LIB_PATH=$PWD/danet-tensorflow_amd/csrc/libdanet_hip_cha$n.so b cha$n; done
done
