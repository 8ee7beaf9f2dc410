cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_sharding_gloo.py -m gpu -q --timeout 600 > gpurun_out/r2_pytest14.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert" gpurun_out/r2_pytest14.log | head -20
