cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash profiles/run_r05.sh r05_k c03156c > gpurun_out/r05_k.log 2>&1
tail -40 gpurun_out/r05_k.log
