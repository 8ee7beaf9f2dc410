cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 --timeout 600 > gpurun_out/r2_pytest8.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert" gpurun_out/r2_pytest8.log | head -20
timeout 600 python bench.py > gpurun_out/r2_bench8.log 2>&1; echo "bench rc=$?"
grep -v amdgpu gpurun_out/r2_bench8.log | tail -3
