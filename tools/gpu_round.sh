cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q --maxfail=10 --timeout 600 -k "one_call or ragged or full_size or oracle" > gpurun_out/r2_pytest7.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert" gpurun_out/r2_pytest7.log | head -10
timeout 600 python tools/bench_fused.py 0 4 > gpurun_out/r2_bench_fused8.log 2>&1; echo "bench_fused rc=$?"
grep -v amdgpu gpurun_out/r2_bench_fused8.log | tail -14
