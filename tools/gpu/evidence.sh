cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash profiles/run_r05.sh r05_j 8e7a4cd > gpurun_out/r05_j.log 2>&1
tail -40 gpurun_out/r05_j.log
