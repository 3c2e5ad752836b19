cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python tools/feed_soak.py 2>&1 | grep -v amdgpu | tail -8 | tee gpurun_out/r05_n_feed_soak.txt
for i in 1 2; do timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -1; done
