cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python tools/gpu/ab_plan.py 2>&1 | grep -v amdgpu.ids
