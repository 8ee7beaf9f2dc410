cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_c_host.py -m gpu -q --timeout 600 > gpurun_out/r2_pytest11.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert" gpurun_out/r2_pytest11.log | head -20
timeout 600 python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu | tail -3
