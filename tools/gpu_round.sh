cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for i in 1 2; do
timeout 1500 python -m pytest tests -m gpu -q --maxfail=10 --timeout 600 -p no:cacheprovider > gpurun_out/r2_pytest_final$i.log 2>&1; echo "pytest rc=$?"
grep -E "passed|failed" gpurun_out/r2_pytest_final$i.log | tail -2
done
timeout 600 python bench.py --cpu-rays 0 > gpurun_out/r2_bench_last.log 2>&1; echo "bench rc=$?"
grep -v amdgpu gpurun_out/r2_bench_last.log | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
