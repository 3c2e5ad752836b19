cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 300 python tools/xcd_local_probe.py 2>&1 | grep -v amdgpu.ids
