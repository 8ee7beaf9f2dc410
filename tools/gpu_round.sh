cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q --maxfail=10 --timeout 600 -k "texel_cache" > gpurun_out/r2_pytest6.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert" gpurun_out/r2_pytest6.log | head -10
timeout 600 python tools/bench_fused.py 0 > gpurun_out/r2_bench_fused6.log 2>&1; echo "bench_fused rc=$?"
grep -v amdgpu gpurun_out/r2_bench_fused6.log | grep -v "TEX ABL" | tail -6
