cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q --timeout 600 -k "common_lattice or one_call_route" > gpurun_out/r2_pytest12.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert" gpurun_out/r2_pytest12.log | head -20
