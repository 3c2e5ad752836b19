# usage (through grun.sh): TAG is baked in below -- edit, or export EVIDENCE_TAG before gpurun snapshots the tree
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${EVIDENCE_TAG:-r06_k}
REV=$(cat tools/gpu/evidence_rev.txt 2>/dev/null || echo unknown)
bash profiles/run_r06.sh $TAG $REV > gpurun_out/$TAG.log 2>&1
tail -40 gpurun_out/$TAG.log
