cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
echo "== product"; timeout 300 python tools/follow_probe.py 2>&1 | tail -9
echo "== slices of 8"; timeout 300 python tools/follow_probe.py 8 2>&1 | tail -3
echo "== plain loads in the followers (timing only)"
DANET_LIB_PATH=$PWD/danet-tensorflow_amd/csrc/libdanet_hip_plainld.so timeout 300 python tools/follow_probe.py 2>&1 | tail -9
