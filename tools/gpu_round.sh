cd $GRAFT_REPO_ROOT
python tools/bw_probe.py 2>&1 | grep -v amdgpu | tail -4
