set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_hip_parity.py -m gpu -q --maxfail=10 --timeout 600 -k "texel_cache or get_z" -s > gpurun_out/r2_pytest4.log 2>&1; echo "pytest rc=$?"
grep -E "handed back|passed|failed|Error|assert" gpurun_out/r2_pytest4.log | head -30
timeout 600 python tools/bench_fused.py 0 > gpurun_out/r2_bench_fused2.log 2>&1; echo "bench_fused rc=$?"
tail -6 gpurun_out/r2_bench_fused2.log
