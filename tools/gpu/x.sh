cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
bash tools/heads_stats.sh > gpurun_out/r05_m_heads_stats.txt 2>&1
bash tools/heads_pmc.sh > gpurun_out/r05_m_heads_pmc.txt 2>&1
tail -30 gpurun_out/r05_m_heads_pmc.txt; cat gpurun_out/r05_m_heads_stats.txt
