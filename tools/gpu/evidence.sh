cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash profiles/run_r05.sh r05_h $(git rev-parse --short HEAD 2>/dev/null || echo unknown) > gpurun_out/r05_h.log 2>&1
tail -40 gpurun_out/r05_h.log
