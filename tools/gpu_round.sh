cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 --timeout 600 > gpurun_out/r2_pytest10.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert" gpurun_out/r2_pytest10.log | head -20
timeout 600 python bench.py > gpurun_out/r2_bench10.log 2>&1; echo "bench rc=$?"
grep -v amdgpu gpurun_out/r2_bench10.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'], d['roofline']['frac'], d['hbm'])"
