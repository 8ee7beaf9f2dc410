cd $GRAFT_REPO_ROOT
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -6
