cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/bench_fused.py 0 16 17 0 > gpurun_out/r2_bench_fused14.log 2>&1; echo "bench_fused rc=$?"
grep -v amdgpu gpurun_out/r2_bench_fused14.log | tail -4
