cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q --maxfail=12 --timeout 600 > gpurun_out/r2_pytest16.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed|Error|assert" gpurun_out/r2_pytest16.log | head -24
timeout 600 python bench.py --cpu-rays 0 --no-extras > gpurun_out/r2_bench16.log 2>&1; echo "bench rc=$?"
grep -v amdgpu gpurun_out/r2_bench16.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'])"
timeout 600 python bench.py --cpu-rays 0 --no-extras --chunk-rays 8192 > gpurun_out/r2_bench16b.log 2>&1; echo "bench rc=$?"
grep -v amdgpu gpurun_out/r2_bench16b.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['stage_ms'])"
