cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 --timeout 600 > gpurun_out/r2_pytest9.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert" gpurun_out/r2_pytest9.log | head -20
timeout 600 python bench.py > gpurun_out/r2_bench9.log 2>&1; echo "bench rc=$?"
grep -v amdgpu gpurun_out/r2_bench9.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'])"
timeout 600 python tools/bench_fused.py 0 1 5 4 > gpurun_out/r2_bench_fused9.log 2>&1; echo "bench_fused rc=$?"
grep -v amdgpu gpurun_out/r2_bench_fused9.log | tail -16
