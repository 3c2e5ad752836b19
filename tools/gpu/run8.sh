cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash profiles/run_r05.sh r05_h $(cat tools/gpu/rev.txt) > gpurun_out/r05_h.log 2>&1
tail -40 gpurun_out/r05_h.log
