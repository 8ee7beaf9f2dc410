cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 600 python tools/bench_fused.py 5 6 7 > gpurun_out/r2_bench_fused16.log 2>&1; echo "bench_fused rc=$?"
grep -v amdgpu gpurun_out/r2_bench_fused16.log | tail -3
