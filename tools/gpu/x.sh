cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_heads.py tests/test_gpu_parity.py tests/test_gpu_trained_parity.py -x -q 2>&1 | tail -3
bash tools/heads_stats.sh 2>&1 | grep "anchor_fwd\|anchor_final"
bash tools/heads_stats.sh --cfg4 2>&1 | grep "anchor_fwd\|truth"
